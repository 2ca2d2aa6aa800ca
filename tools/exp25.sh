timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "context_table or english_like or random_indexes or golden or text96" --tb=short 2>&1 | tail -4 | cut -c1-300
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 --pmc off --steps 10 --warmup 3 --workload eng "$@" > gpurun_out/exp25_$tag.json 2> gpurun_out/exp25_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp25_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms locate %.3f compulsory %.2f GB open %.1f tableGB %.1f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r.get("locate_kernel_ms") or 0, r["compulsory_bytes_per_launch"]/1e9, d["config"]["open_s"], d["config"]["index"]["table_bytes"]/1e9), d["config"]["index"]["packed_lines"].get("context2_syms"), {k:v for k,v in r["compulsory"]["count"]["distinct_lines"].items() if v})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp25_$tag.err").read()[-1500:])
PY
}
run h12
FEMTO_AMD_CTX2_SYMS=11 run h11
FEMTO_AMD_CTX2_SYMS=14 run h14
FEMTO_AMD_CTX2=0 run off
