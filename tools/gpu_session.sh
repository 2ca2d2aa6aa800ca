set -u
mkdir -p gpurun_out/s18
SECONDS=0
python bench.py --steps 20 --warmup 5 > gpurun_out/s18/bench_default.json 2> gpurun_out/s18/bench_default.err
echo "default bench wall: $SECONDS s"; tail -2 gpurun_out/s18/bench_default.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/s18/bench_default.json'))
r=d['roofline']
print('headline', round(d['value']/1e9,2), 'G/s', round(d['ms_per_step'],3), 'ms count', round(r['count_kernel_ms'],3), 'loc', round(r['locate_kernel_ms'],3), 'frac', round(r['frac'],3), 'useful', round(r['useful']['frac'],3), 'traffic/comp', r['traffic_over_compulsory'])
for k,v in d['extra'].items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','ms_per_step','ms','count_kernel_ms','locate_kernel_ms','error','equal_to_symbol_path','cpu_baseline')})
e=d['extra']['cfg3_text96_count_locate'].get('roofline')
if e: print('cfg3 roofline frac', round(e['frac'],3), 'useful', round(e['useful']['frac'],3), 'traffic/comp', e['traffic_over_compulsory'])
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_integration.py -m gpu -x -q -k "not full_size" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | cut -c1-200
echo "all: $SECONDS s"
