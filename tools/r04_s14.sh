set -u
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size or text96" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" | tail -4
echo "tests: $SECONDS s"
for i in 1 2; do
timeout 200 bash tools/quick_bench.sh "eng hint" -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 200 bash tools/quick_bench.sh "eng head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 10 --warmup 3 --pmc off --workload eng
done
echo "all: $SECONDS s"
