# err-flag isolation test, fuzz with repro data, host-pointer path with the spinning pool
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "invalid_pattern or concurrent or device or random_indexes" --tb=short 2>&1 | tail -5 | cut -c1-300
mkdir -p gpurun_out/fuzzbad
FUZZ_KEEP=gpurun_out/fuzzbad timeout 900 python tools/fuzz_gpu.py 1 140 > gpurun_out/exp16_fuzz.log 2>&1; tail -3 gpurun_out/exp16_fuzz.log | cut -c1-600
FUZZ_KEEP=gpurun_out/fuzzbad timeout 900 python tools/fuzz_gpu.py 2 140 > gpurun_out/exp16_fuzz2.log 2>&1; tail -3 gpurun_out/exp16_fuzz2.log | cut -c1-600
python bench.py --steps 3 --warmup 1 --no-extra > /dev/null 2>&1   # builds the bench index
for c in 21 20 19; do FEMTO_AMD_PIPE_CHUNK_LOG2=$c python tools/host_path_bench.py 2>&1 | tail -1; done
for t in 32 64 96; do FEMTO_AMD_HOST_THREADS=$t FEMTO_AMD_PIPE_CHUNK_LOG2=20 python tools/host_path_bench.py 2>&1 | tail -1; done
