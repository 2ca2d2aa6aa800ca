set -u
SECONDS=0
bash tools/profile_round.sh r04_default --steps 20 --warmup 5 > gpurun_out/r04_default.log 2>&1; tail -1 gpurun_out/r04_default.log | cut -c1-100; echo "default $SECONDS s"
bash tools/profile_round.sh r04_eng --steps 10 --warmup 3 --workload eng --no-extra > gpurun_out/r04_eng.log 2>&1; tail -1 gpurun_out/r04_eng.log | cut -c1-100; echo "eng $SECONDS s"
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04/s15_tests.log 2>&1
grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" gpurun_out/r04/s15_tests.log | tail -4 | cut -c1-300
echo "all: $SECONDS s"
