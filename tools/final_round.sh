# tools/final_round.sh -- on the GPU box (gpurun): the whole -m gpu suite, then the profile rounds that profiles/ is promoted from
# (tools/promote_profiles.py r02_default r02_eng r02_hit; the reads100 / cfg5 bench lines are copied as they are).
timeout 2700 python -m pytest tests -m gpu -x -q -rs 2>&1 | grep -E "passed|failed|Error|error|assert|SKIPPED" | tail -12 | cut -c1-200
bash tools/profile_round.sh r02_default --steps 20 --warmup 5 > gpurun_out/r02_default.log 2>&1; tail -2 gpurun_out/r02_default.log | cut -c1-200
bash tools/profile_round.sh r02_eng --steps 10 --warmup 3 --workload eng --no-extra > gpurun_out/r02_eng.log 2>&1; tail -2 gpurun_out/r02_eng.log | cut -c1-200
bash tools/profile_round.sh r02_hit --steps 10 --warmup 3 --workload acgt_hit --no-extra > gpurun_out/r02_hit.log 2>&1; tail -2 gpurun_out/r02_hit.log | cut -c1-200
python bench.py --steps 10 --warmup 3 --workload acgt_hit --plen 100 --npats 4000000 --no-extra > gpurun_out/r02_reads100_bench.json 2> gpurun_out/r02_reads100.err; tail -1 gpurun_out/r02_reads100.err
python bench.py --steps 10 --warmup 3 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 20000 --ref-sample 10000 > gpurun_out/r02_cfg5_8GiB_bench.json 2> gpurun_out/r02_cfg5_8GiB.err; tail -2 gpurun_out/r02_cfg5_8GiB.err | cut -c1-300
for i in 1 2 3; do python tools/host_path_bench.py 2>&1 | tail -1; done
