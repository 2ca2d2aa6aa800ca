timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "context_table or english_like or random_indexes or golden or long_pattern or invalid" --tb=short 2>&1 | tail -6 | cut -c1-400
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 "$@" > gpurun_out/exp19_$tag.json 2> gpurun_out/exp19_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp19_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms frac %.3f traffic %.2f GB compulsory %.2f GB open %.1fs" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("traffic") or 0)/1e9, r["compulsory_bytes_per_launch"]/1e9, d["config"]["open_s"]), d["config"]["index"]["packed_lines"], "tableGB %.1f" % (d["config"]["index"]["table_bytes"]/1e9))
    print("   ", {k:v for k,v in r["compulsory"]["count"]["distinct_lines"].items() if v})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp19_$tag.err").read()[-1500:])
PY
}
run eng --steps 10 --warmup 3 --workload eng
FEMTO_AMD_CTX=0 run eng_noctx --steps 10 --warmup 3 --workload eng --pmc off
FEMTO_AMD_CTX_SYMS=7 run eng_h7 --steps 10 --warmup 3 --workload eng --pmc off
FEMTO_AMD_CTX_SYMS=6 run eng_h6 --steps 10 --warmup 3 --workload eng --pmc off
run default --steps 20 --warmup 5
run hit --steps 10 --warmup 3 --workload acgt_hit --pmc off
