set -u
SECONDS=0
B=4294967296
for v in "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=1" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=1" "FEMTO_AMD_PIPE_LAG=1 FEMTO_AMD_NT_STORES=0" "FEMTO_AMD_PIPE_LAG=2 FEMTO_AMD_NT_STORES=0"; do
  echo "== $v"; env $v python tools/host_path_bench.py 2>&1 | tail -2
done
echo "host: $SECONDS s"
for i in 1 2; do
timeout 90 bash tools/quick_bench.sh "budget hit ru" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget hit noru" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B,rank_units=0
timeout 90 bash tools/quick_bench.sh "budget rand ru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 90 bash tools/quick_bench.sh "budget rand noru" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B,rank_units=0
timeout 90 bash tools/quick_bench.sh "default hit ru" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit
timeout 90 bash tools/quick_bench.sh "default hit noru" -- --steps 10 --warmup 3 --pmc off --workload acgt_hit --open-opts rank_units=0
timeout 90 bash tools/quick_bench.sh "default rand ru" -- --steps 20 --warmup 5 --pmc off
timeout 90 bash tools/quick_bench.sh "default rand noru" -- --steps 20 --warmup 5 --pmc off --open-opts rank_units=0
done
echo "all: $SECONDS s"
