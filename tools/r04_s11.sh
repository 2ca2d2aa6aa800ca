set -u
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" | tail -4
echo "tests: $SECONDS s"
for i in 1 2; do
timeout 200 bash tools/quick_bench.sh "eng new" -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 200 bash tools/quick_bench.sh "eng head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 10 --warmup 3 --pmc off --workload eng
timeout 90 bash tools/quick_bench.sh "hit new" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "hit head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "reads100 new" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --plen 100 --npats 4000000
timeout 90 bash tools/quick_bench.sh "reads100 head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --plen 100 --npats 4000000
timeout 90 bash tools/quick_bench.sh "default new" -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "default head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 20 --warmup 5 --pmc off
done
echo "all: $SECONDS s"
