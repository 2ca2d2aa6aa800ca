FUZZ_RUN=1 timeout 300 python tools/fuzz_gpu.py 1 41 40 2>&1 | tail -2 | cut -c1-300
timeout 900 python tools/fuzz_gpu.py 3 140 > gpurun_out/exp18_fuzz3.log 2>&1; tail -2 gpurun_out/exp18_fuzz3.log | cut -c1-400
run() { tag=$1; shift; python bench.py --no-extra --cpu-sample 0 "$@" > gpurun_out/exp18_$tag.json 2> gpurun_out/exp18_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp18_$tag.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$tag", "%.3f G/s %.3f ms kernel %.3f ms frac %.3f traffic %.2f GB compulsory %.2f GB open %.1fs" % (d["value"]/1e9, d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("traffic") or 0)/1e9, r["compulsory_bytes_per_launch"]/1e9, d["config"]["open_s"]), d["config"]["index"]["packed_lines"])
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/exp18_$tag.err").read()[-800:])
PY
}
run default --steps 20 --warmup 5 --pmc off
FEMTO_AMD_KTAB_SYMS=16 FEMTO_AMD_KTAB_MB=80000 run default_k16 --steps 20 --warmup 5
FEMTO_AMD_KTAB_SYMS=16 FEMTO_AMD_KTAB_MB=80000 run hit_k16 --steps 10 --warmup 3 --workload acgt_hit --pmc off
for i in 1 2; do for c in 20 21; do FEMTO_AMD_PIPE_CHUNK_LOG2=$c python tools/host_path_bench.py 2>&1 | tail -1; done; done
run cfg5_8GiB --steps 5 --warmup 2 --workload acgt_hit --text-log2 33
