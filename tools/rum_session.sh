# tools/rum_session.sh -- on the GPU box: the marked rank units against the plain ones on the footprint-bounded handle (same box, alternating)
set -u
SECONDS=0
B=4294967296
OLD="FEMTO_AMD_RU=2 FEMTO_AMD_WAVELET_LINES=1 FEMTO_AMD_MARK_EVERY=10"
for rep in 1 2; do
timeout 200 bash tools/quick_bench.sh "hit  marked" -- --steps 10 --warmup 3 --workload acgt_hit --pmc off --open-opts hbm_budget_bytes=$B
timeout 200 bash tools/quick_bench.sh "hit  plain " $OLD -- --steps 10 --warmup 3 --workload acgt_hit --pmc off --open-opts hbm_budget_bytes=$B
timeout 200 bash tools/quick_bench.sh "rand marked" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
timeout 200 bash tools/quick_bench.sh "rand plain " $OLD -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
done
timeout 200 bash tools/quick_bench.sh "hit  marked, marks 10" FEMTO_AMD_MARK_EVERY=10 -- --steps 10 --warmup 3 --workload acgt_hit --pmc off --open-opts hbm_budget_bytes=$B
timeout 200 bash tools/quick_bench.sh "hit  default bound (8x)" -- --steps 10 --warmup 3 --workload acgt_hit --pmc off --open-opts hbm_budget_bytes=-1
timeout 200 bash tools/quick_bench.sh "rand default bound (8x)" -- --steps 20 --warmup 5 --pmc off --open-opts hbm_budget_bytes=-1
echo "all $SECONDS s"
