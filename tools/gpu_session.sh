set -u
SECONDS=0
export TMPDIR=/tmp
O=gpurun_out/r06_regexp; mkdir -p $O
for T in 4 8; do
  python tools/regexp_bench.py --which approx --reps 2 --concurrent $T 2>/dev/null | grep -v nodes_avg > $O/concurrent_$T.json
  grep concurrent $O/concurrent_$T.json | cut -c1-400
done
echo "regexp $SECONDS s"
bash tools/budget_sweep_eng.sh > gpurun_out/r06_budget_sweep_eng.log 2>&1
cat gpurun_out/r06_budget_sweep_eng.log | cut -c1-260
echo "eng sweep $SECONDS s"
bash tools/budget_sweep.sh > gpurun_out/r06_budget_sweep.log 2>&1
cat gpurun_out/r06_budget_sweep.log | cut -c1-260
echo "all $SECONDS s"
