set -u
SECONDS=0
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_regexp.py -x -q -m gpu 2>&1 | tail -6
echo "t1 $SECONDS s"
for b in -1; do
  B=-1; L="default (8x)"
  timeout 400 bash tools/quick_bench.sh "eng budget $L" -- --steps 5 --warmup 2 --pmc off --workload eng --open-opts hbm_budget_bytes=$B
  timeout 400 bash tools/quick_bench.sh "eng budget $L row-free" -- --steps 5 --warmup 2 --pmc off --workload eng --row-free --open-opts hbm_budget_bytes=$B
done
echo "t3 $SECONDS s"
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "text96" 2>&1 | tail -12
echo "t4 $SECONDS s"
python tools/exchange_bench.py --npats 500000 2>gpurun_out/exch.err | cut -c1-700; tail -3 gpurun_out/exch.err
echo "all $SECONDS s"
