set -u
SECONDS=0
timeout 760 python bench.py --steps 10 --warmup 3 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 20000 --ref-sample 10000 > gpurun_out/r03_cfg5_rerun.json 2> gpurun_out/r03_cfg5_rerun.err; grep -v amdgpu.ids gpurun_out/r03_cfg5_rerun.err | tail -2 | cut -c1-300
echo "all: $SECONDS s"
