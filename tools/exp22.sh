timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "context_table or random_indexes or golden or invalid or million or device or concurrent or locate_range or max_occs or pointer_array or striped or multi_device" --tb=short 2>&1 | tail -4 | cut -c1-300
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 "$@" > gpurun_out/exp22_$tag.json 2> gpurun_out/exp22_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp22_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms locate %.3f frac %.3f traffic %.2f GB compulsory %.2f GB" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r.get("locate_kernel_ms") or 0, r["frac"], (r.get("traffic") or 0)/1e9, r["compulsory_bytes_per_launch"]/1e9))
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp22_$tag.err").read()[-1500:])
PY
}
run default --steps 20 --warmup 5 --pmc off
run eng --steps 10 --warmup 3 --workload eng --pmc off
run hit --steps 10 --warmup 3 --workload acgt_hit --pmc off
run reads100 --steps 10 --warmup 3 --workload acgt_hit --plen 100 --npats 4000000 --pmc off
