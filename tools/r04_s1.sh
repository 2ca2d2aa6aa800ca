set -u
SECONDS=0
mkdir -p gpurun_out/r04
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5
echo "tests: $SECONDS s"
B=4294967296
timeout 120 bash tools/quick_bench.sh "default ru" -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "default noru" -- --steps 20 --warmup 5 --pmc off --open-opts rank_units=0
timeout 90 bash tools/quick_bench.sh "budget4x ru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget4x noru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,rank_units=0
timeout 90 bash tools/quick_bench.sh "budget4x ru K13" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,level_table_syms=13
timeout 90 bash tools/quick_bench.sh "budget4x noru K13" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,level_table_syms=13,rank_units=0
timeout 90 bash tools/quick_bench.sh "budget4x ru K12" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,level_table_syms=12
timeout 90 bash tools/quick_bench.sh "budget4x ru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget4x noru hit" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B,rank_units=0
echo "all: $SECONDS s"
