set -u
mkdir -p gpurun_out/s9
python bench.py --pmc off --steps 3 --warmup 1 --cpu-sample 0 --no-extra > /dev/null 2>&1   # builds the index
for cfg in "" "FEMTO_AMD_HOST_THREADS_OUT=16" "FEMTO_AMD_HOST_THREADS_OUT=64" "FEMTO_AMD_HOST_THREADS=96 FEMTO_AMD_HOST_THREADS_OUT=32" "FEMTO_AMD_PIPE_CHUNK_LOG2=19" "FEMTO_AMD_PIPE_CHUNK_LOG2=21"; do
  env $cfg python tools/host_path_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a gpurun_out/s9/host.txt
done
python tools/host_locate_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee -a gpurun_out/s9/host.txt
python -m pytest tests/test_gpu_parity.py tests/test_integration.py -x -q -m gpu -k "not full_size" 2>&1 | tail -8 | tee gpurun_out/s9/parity.txt
