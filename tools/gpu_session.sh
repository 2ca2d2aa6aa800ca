set -u
mkdir -p gpurun_out/s16
SECONDS=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/s16/smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q -rs 2>&1 | grep -E "passed|failed|Error|error|assert|SKIPPED" | tail -14 | cut -c1-220 | tee gpurun_out/s16/pytest.txt
echo "tests: $SECONDS s"
for so in 1 0; do
  bash tools/quick_bench.sh mode1_sort$so FEMTO_AMD_RANK_MODE=lane FEMTO_AMD_SORT=$so -- --pmc off --steps 5 --warmup 2 --npats 2000000 --cpu-sample 0 2>&1 | tee -a gpurun_out/s16/mode1.txt
  bash tools/quick_bench.sh mode1_hit_sort$so FEMTO_AMD_RANK_MODE=lane FEMTO_AMD_SORT=$so -- --pmc off --steps 5 --warmup 2 --npats 2000000 --workload acgt_hit --cpu-sample 0 2>&1 | tee -a gpurun_out/s16/mode1.txt
done
echo "all: $SECONDS s"
