set -u
mkdir -p gpurun_out/s3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size and not random_indexes" 2>&1 | tail -15 | tee gpurun_out/s3/parity.txt
python bench.py --pmc off --steps 20 --warmup 5 --cpu-sample 20000 > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err
tail -3 gpurun_out/s3/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s3/bench.json'))
print('headline', d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('count_kernel_ms'))
for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','ms','count_kernel_ms','locate_kernel_ms','error','equal_to_symbol_path')})
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -f csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --pmc off --steps 6 --warmup 2 --cpu-sample 0 --no-extra > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/trace_gaps.py /tmp/kt 2 | tee gpurun_out/s3/gaps.txt
