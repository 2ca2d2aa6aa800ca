set -u
SECONDS=0
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_free or device_chain" 2>&1 | tail -8
echo "t1 $SECONDS s"
python -m pytest tests/test_gpu_built.py -x -q -m gpu -k "long_patterns" 2>&1 | tail -8
echo "t2 $SECONDS s"
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "text96 or 1gib" 2>&1 | tail -15
echo "t3 $SECONDS s"
python -m pytest tests/test_regexp.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
echo "t4 $SECONDS s"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -c 3000 gpurun_out/bench_a.err
python - <<'PY'
import json
for line in open('gpurun_out/bench_a.json'):
    try: d=json.loads(line)
    except Exception: continue
    if 'extra' in d:
        if d['extra'] in ('p_hit_count_locate','cfg3_text96_count_locate'):
            print(d['extra'], d.get('value'), d.get('ms_per_step'), 'cnt', d.get('count_kernel_ms'), 'loc', d.get('locate_kernel_ms'), 'frac', (d.get('roofline') or {}).get('frac'))
            rf=d.get('row_free') or {}
            print('   row_free', rf.get('value'), rf.get('ms_per_step'), 'cnt', rf.get('count_kernel_ms'), 'loc', rf.get('locate_kernel_ms'), 'frac', (rf.get('roofline') or {}).get('frac'), rf.get('error'))
            do=d.get('default_open') or {}
            if do: print('   default_open', do.get('value'), do.get('ms_per_step'), 'cnt', do.get('count_kernel_ms'), 'loc', do.get('locate_kernel_ms'), do.get('error'), 'rf', (do.get('row_free') or {}).get('value'), (do.get('row_free') or {}).get('count_kernel_ms'), (do.get('row_free') or {}).get('locate_kernel_ms'))
        else:
            print(d['extra'], d.get('value'), d.get('error'))
    else:
        print('HEAD', d['value'], d['ms_per_step'], {k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))})
PY
echo "all $SECONDS s"
