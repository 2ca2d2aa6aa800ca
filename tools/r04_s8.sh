set -u
SECONDS=0
mkdir -p gpurun_out/r04
( time python bench.py > gpurun_out/r04/bench_full.out 2> gpurun_out/r04/bench_full.err ) 2>&1 | tail -3
tail -3 gpurun_out/r04/bench_full.err | cut -c1-300
python - <<'PY'
import json
for ln in open('gpurun_out/r04/bench_full.out'):
    ln=ln.strip()
    if not ln.startswith('{'): continue
    d=json.loads(ln)
    if 'extra' in d:
        print('EXTRA', d['extra'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','ms_per_step','ms','error','count_kernel_ms','equal_to_headline_results','equal_to_device_path')}, 'roof' , {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('roofline') or {}).items() if k in ('frac','kernel_ms','traffic_GBs','achieved')}, len(ln))
    else:
        r=d['roofline']
        print('HEADLINE', round(d['value']/1e9,2), round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'reads', round(r['line_reads']['frac'],3), 'traffic', r.get('traffic_GBs'), 'len', len(ln))
        print(' budget4x', r.get('budget4x')); print(' mode1', r.get('mode1')); print(' reffmt', r.get('reference_format'))
        print(' extras', d.get('extras'))
        print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
echo "bench: $SECONDS s"
for v in "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=1" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=1" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=0"; do
  echo "== $v"; env $v python tools/host_path_bench.py 2>&1 | tail -2
done
echo "host: $SECONDS s"
timeout 200 bash tools/quick_bench.sh "eng" -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 200 bash tools/quick_bench.sh "eng" -- --steps 10 --warmup 3 --pmc off --workload eng
echo "all: $SECONDS s"
