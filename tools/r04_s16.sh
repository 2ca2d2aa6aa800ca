python bench.py --steps 3 --warmup 1 --no-extra --pmc off --cpu-sample 0 > /dev/null 2>&1
for i in 1 2; do echo "== default"; REPS=21 python tools/host_path_bench.py 2>&1 | tail -3; done
echo "== 24"; FEMTO_AMD_HOST_THREADS=24 REPS=21 python tools/host_path_bench.py 2>&1 | tail -3
echo "== 40"; FEMTO_AMD_HOST_THREADS=40 REPS=21 python tools/host_path_bench.py 2>&1 | tail -3
