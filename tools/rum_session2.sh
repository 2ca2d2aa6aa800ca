# tools/rum_session2.sh -- on the GPU box: random patterns on the default bound, plain against marked rank units (same box, alternating)
set -u
SECONDS=0
for rep in 1 2; do
timeout 200 bash tools/quick_bench.sh "rand 8x marked" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=-1
timeout 200 bash tools/quick_bench.sh "rand 8x plain " FEMTO_AMD_RU=2 -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=-1
timeout 200 bash tools/quick_bench.sh "rand 8x plain, lines kept" FEMTO_AMD_RU=2 FEMTO_AMD_WAVELET_LINES=1 -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=-1
done
timeout 200 bash tools/quick_bench.sh "hit 8x plain " FEMTO_AMD_RU=2 -- --steps 10 --warmup 3 --workload acgt_hit --pmc off --open-opts hbm_budget_bytes=-1
echo "all $SECONDS s"
