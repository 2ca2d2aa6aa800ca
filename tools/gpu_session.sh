set -u
mkdir -p gpurun_out/s15
python bench.py --pmc off --steps 3 --warmup 1 --cpu-sample 0 --no-extra > /dev/null 2>&1
python tools/regexp_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/s15/regexp.txt
# mode 1 (femto's wavelet tree) with and without the suffix-order batch sort
for so in 1 0; do
  bash tools/quick_bench.sh mode1_sort$so FEMTO_AMD_RANK_MODE=lane FEMTO_AMD_SORT=$so -- --pmc off --steps 5 --warmup 2 --npats 2000000 2>&1 | tee -a gpurun_out/s15/mode1.txt
done
