set -u
mkdir -p gpurun_out/s11
SECONDS=0
bash tools/profile_round.sh r03_default --steps 20 --warmup 5 > gpurun_out/r03_default.log 2>&1; tail -2 gpurun_out/r03_default.log | cut -c1-200
echo "default round: $SECONDS s"
bash tools/profile_round.sh r03_eng --steps 10 --warmup 3 --workload eng --no-extra > gpurun_out/r03_eng.log 2>&1; tail -2 gpurun_out/r03_eng.log | cut -c1-200
bash tools/profile_round.sh r03_hit --steps 10 --warmup 3 --workload acgt_hit --no-extra > gpurun_out/r03_hit.log 2>&1; tail -2 gpurun_out/r03_hit.log | cut -c1-200
python bench.py --steps 10 --warmup 3 --workload acgt_hit --plen 100 --npats 4000000 --no-extra > gpurun_out/r03_reads100_bench.json 2> gpurun_out/r03_reads100.err; tail -1 gpurun_out/r03_reads100.err
echo "rounds: $SECONDS s"
for k in 12 13 14 15 16; do
  bash tools/quick_bench.sh K$k FEMTO_AMD_KTAB_SYMS=$k -- --pmc off --steps 20 --warmup 5 2>&1 | tee -a gpurun_out/s11/ktab_sweep.txt
  python - <<PY | tee -a gpurun_out/s11/ktab_sweep.txt
import os, sys
sys.path.insert(0, '.')
os.environ['FEMTO_AMD_KTAB_SYMS'] = '$k'
PY
done
echo "sweep: $SECONDS s"
python bench.py --steps 3 --warmup 1 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 0 --pmc off > /dev/null 2>&1
TEXT_LOG2=33 WORKLOAD=hit python tools/striped_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/s11/striped_8gib.txt
echo "all: $SECONDS s"
