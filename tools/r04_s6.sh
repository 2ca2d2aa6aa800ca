set -u
SECONDS=0
B=4294967296
A="--workload acgt_hit --open-opts hbm_budget_bytes=$B"
bash tools/pmc_only.sh r04_hit_ru TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- $A 2>&1 | grep -A12 "count_direct" | head -30
bash tools/pmc_only.sh r04_hit_noru TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- $A,rank_units=0 2>&1 | grep -A12 "count_direct" | head -30
echo "pmc: $SECONDS s"
for i in 1 2; do
timeout 90 bash tools/quick_bench.sh "budget hit ru8" -- --steps 10 --warmup 3 --pmc off $A
timeout 90 bash tools/quick_bench.sh "budget hit ru7" FEMTO_AMD_LIB=$PWD/ab/lib_ru7.so -- --steps 10 --warmup 3 --pmc off $A
timeout 90 bash tools/quick_bench.sh "budget rand ru8" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget rand ru7" FEMTO_AMD_LIB=$PWD/ab/lib_ru7.so -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
done
python tools/host_path_bench.py 2>&1 | tail -3
python tools/host_path_bench.py 2>&1 | tail -1
echo "all: $SECONDS s"
