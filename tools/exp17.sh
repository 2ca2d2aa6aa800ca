# LDS pattern chunks + tail rule: parity subset, then the four workloads with live traffic
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or random_indexes or invalid or device or concurrent or long_pattern or tail" --tb=short 2>&1 | tail -4 | cut -c1-300
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 "$@" > gpurun_out/exp17_$tag.json 2> gpurun_out/exp17_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/exp17_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$tag", "%.3f G/s %.3f ms kernel %.3f ms frac %.3f traffic %.2f GB compulsory %.2f GB" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("traffic") or 0)/1e9, r["compulsory_bytes_per_launch"]/1e9))
PY
}
run default --steps 20 --warmup 5
FEMTO_AMD_TAIL_ONES=1 run default_ones1 --steps 20 --warmup 5 --pmc off
FEMTO_AMD_TAIL_ONES=0 run default_ones0 --steps 20 --warmup 5 --pmc off
run hit --steps 10 --warmup 3 --workload acgt_hit
FEMTO_AMD_TAIL_ONES=1 run hit_ones1 --steps 10 --warmup 3 --workload acgt_hit --pmc off
FEMTO_AMD_TAIL_ONES=0 run hit_ones0 --steps 10 --warmup 3 --workload acgt_hit --pmc off
run reads100 --steps 10 --warmup 3 --workload acgt_hit --plen 100 --npats 4000000
run eng --steps 10 --warmup 3 --workload eng
FUZZ_RUN=1 timeout 300 python tools/fuzz_gpu.py 1 41 40 2>&1 | tail -2 | cut -c1-300
python tools/host_path_bench.py 2>&1 | tail -1
