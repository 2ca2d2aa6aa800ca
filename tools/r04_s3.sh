set -u
SECONDS=0
mkdir -p gpurun_out/r04
B=4294967296
bash tools/profile_round.sh r04_budget_ru --no-extra --steps 20 --warmup 5 --open-opts hbm_budget_bytes=$B
echo "profile ru: $SECONDS s"
bash tools/profile_round.sh r04_budget_noru --no-extra --steps 20 --warmup 5 --open-opts hbm_budget_bytes=$B,rank_units=0
echo "all: $SECONDS s"
