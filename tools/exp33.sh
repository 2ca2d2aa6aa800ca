timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or random_indexes or long_pattern or gpu_built or million or context_table" --tb=short 2>&1 | tail -3 | cut -c1-300
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 --pmc off "$@" > gpurun_out/exp33_$tag.json 2> gpurun_out/exp33_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp33_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms locate %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r.get("locate_kernel_ms") or 0))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp33_$tag.err").read()[-1500:])
PY
}
run default --steps 20 --warmup 5
run hit --steps 10 --warmup 3 --workload acgt_hit
run reads100 --steps 10 --warmup 3 --workload acgt_hit --plen 100 --npats 4000000
