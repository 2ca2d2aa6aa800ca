set -u
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" | tail -4
echo "tests: $SECONDS s"
for i in 1 2; do
timeout 200 bash tools/quick_bench.sh "eng spec" -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 200 bash tools/quick_bench.sh "eng nospec" FEMTO_AMD_TAIL_SPEC=0 -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 90 bash tools/quick_bench.sh "hit spec" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "hit nospec" FEMTO_AMD_TAIL_SPEC=0 -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "default spec" -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "default nospec" FEMTO_AMD_TAIL_SPEC=0 -- --steps 20 --warmup 5 --pmc off
done
echo "all: $SECONDS s"
