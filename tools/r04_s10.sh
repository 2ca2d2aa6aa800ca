set -u
SECONDS=0
B=4294967296
for i in 1 2; do
timeout 120 bash tools/quick_bench.sh "default persistent" -- --steps 20 --warmup 5 --pmc off
timeout 120 bash tools/quick_bench.sh "default head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 20 --warmup 5 --pmc off
timeout 120 bash tools/quick_bench.sh "default persistent-off" FEMTO_AMD_PERSISTENT=0 -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "budget persistent" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "hit persistent" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "hit head" FEMTO_AMD_LIB=$PWD/ab/lib_head.so -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "search_cli or device_chain" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" | tail -5
echo "all: $SECONDS s"
