set -u
SECONDS=0
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_regexp.py -x -q -m gpu -k "not full_size" -p no:cacheprovider > gpurun_out/r04/s4_tests.log 2>&1
grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" gpurun_out/r04/s4_tests.log | tail -4
echo "tests: $SECONDS s"
B=4294967296
for i in 1 2; do
timeout 120 bash tools/quick_bench.sh "budget4x ru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget4x noru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,rank_units=0
timeout 120 bash tools/quick_bench.sh "default ru" -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "default noru" -- --steps 20 --warmup 5 --pmc off --open-opts rank_units=0
done
timeout 90 bash tools/quick_bench.sh "budget4x ru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget4x noru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B,rank_units=0
timeout 90 bash tools/quick_bench.sh "default ru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "default noru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts rank_units=0
echo "all: $SECONDS s"
