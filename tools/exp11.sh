python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra --pmc off > /dev/null 2>&1
for t in 16 32 64 128; do FEMTO_AMD_HOST_THREADS=$t python tools/host_path_bench.py 2>&1 | tail -1; done
for c in 18 19 20; do FEMTO_AMD_PIPE_CHUNK_LOG2=$c python tools/host_path_bench.py 2>&1 | tail -1; done
FEMTO_AMD_HOST_THREADS=32 FEMTO_AMD_PIPE_CHUNK_LOG2=19 python tools/host_path_bench.py 2>&1 | tail -1
FEMTO_AMD_HOST_KEYS=0 python tools/host_path_bench.py 2>&1 | tail -1
