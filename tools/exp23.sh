run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 --pmc off --steps 10 --warmup 3 --workload eng "$@" > gpurun_out/exp23_$tag.json 2> gpurun_out/exp23_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp23_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms locate %.3f compulsory %.2f GB" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r.get("locate_kernel_ms") or 0, r["compulsory_bytes_per_launch"]/1e9), {k:v for k,v in r["compulsory"]["count"]["distinct_lines"].items() if v})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp23_$tag.err").read()[-1500:])
PY
}
FEMTO_AMD_TAIL_ROWS=1 run rows1
FEMTO_AMD_TAIL_ROWS=2 FEMTO_AMD_TAIL_ROW_COST=4 run rows2_c4
FEMTO_AMD_TAIL_ROWS=2 FEMTO_AMD_TAIL_ROW_COST=12 run rows2_c12
FEMTO_AMD_TAIL_ROWS=4 FEMTO_AMD_TAIL_ROW_COST=8 run rows4_c8
FEMTO_AMD_TAIL_ROWS=4 FEMTO_AMD_TAIL_ROW_COST=16 run rows4_c16
