set -u
mkdir -p gpurun_out/s7
SECONDS=0
python bench.py > gpurun_out/s7/bench_default.json 2> gpurun_out/s7/bench_default.err
echo "default bench wall: $SECONDS s"
tail -3 gpurun_out/s7/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s7/bench_default.json'))
r=d['roofline']
print('headline', round(d['value']/1e9,2), 'G/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],3), 'useful', round(r['useful']['frac'],3), 'traffic/comp', r['traffic_over_compulsory'])
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','kind','gpu_vs_cpu')})
for k,v in d['extra'].items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','ms_per_step','ms','count_kernel_ms','locate_kernel_ms','error','equal_to_symbol_path')})
e=d['extra']['cfg3_text96_count_locate'].get('roofline')
if e: print('cfg3 roofline frac', round(e['frac'],3), 'useful', round(e['useful']['frac'],3), 'traffic/comp', e['traffic_over_compulsory'], 'ctx2', d['extra']['cfg3_text96_count_locate']['index'].get('context2_syms'))
PY
