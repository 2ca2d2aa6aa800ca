set -u
SECONDS=0
timeout 900 python tools/fuzz_gpu.py 41 120 2>&1 | tail -6
echo "fuzz: $SECONDS s"
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04/s13_tests.log 2>&1
grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" gpurun_out/r04/s13_tests.log | tail -6 | cut -c1-300
echo "tests: $SECONDS s"
( time python bench.py > gpurun_out/r04/bench_final.out 2> gpurun_out/r04/bench_final.err ) 2>&1 | tail -3
tail -c 1500 gpurun_out/r04/bench_final.out | cut -c1-600
echo "all: $SECONDS s"
