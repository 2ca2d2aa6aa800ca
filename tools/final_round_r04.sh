# tools/final_round_r04.sh -- on the GPU box (gpurun): the profile rounds profiles/r04_* are promoted from
# (python tools/promote_profiles.py r04_default r04_budget4x r04_mode1 r04_cfg5 r04_eng r04_hit afterwards, in the container)
set -u
SECONDS=0
B=4294967296
bash tools/profile_round.sh r04_default --steps 20 --warmup 5 > gpurun_out/r04_default.log 2>&1; tail -2 gpurun_out/r04_default.log | cut -c1-200; echo "default $SECONDS s"
bash tools/profile_round.sh r04_budget4x --steps 20 --warmup 5 --no-extra --open-opts hbm_budget_bytes=$B > gpurun_out/r04_budget4x.log 2>&1; tail -2 gpurun_out/r04_budget4x.log | cut -c1-200; echo "budget $SECONDS s"
bash tools/profile_round.sh r04_budget4x_hit --steps 10 --warmup 3 --no-extra --workload acgt_hit --open-opts hbm_budget_bytes=$B > gpurun_out/r04_budget4x_hit.log 2>&1; tail -2 gpurun_out/r04_budget4x_hit.log | cut -c1-200; echo "budget hit $SECONDS s"
FEMTO_AMD_RANK_MODE=lane bash tools/profile_round.sh r04_mode1 --steps 3 --warmup 1 --no-extra > gpurun_out/r04_mode1.log 2>&1; tail -2 gpurun_out/r04_mode1.log | cut -c1-200; echo "mode1 $SECONDS s"
bash tools/profile_round.sh r04_hit --steps 10 --warmup 3 --workload acgt_hit --no-extra > gpurun_out/r04_hit.log 2>&1; tail -2 gpurun_out/r04_hit.log | cut -c1-200; echo "hit $SECONDS s"
bash tools/profile_round.sh r04_eng --steps 10 --warmup 3 --workload eng --no-extra > gpurun_out/r04_eng.log 2>&1; tail -2 gpurun_out/r04_eng.log | cut -c1-200; echo "eng $SECONDS s"
bash tools/profile_round.sh r04_cfg5 --steps 6 --warmup 2 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 20000 --ref-sample 10000 > gpurun_out/r04_cfg5.log 2>&1; tail -2 gpurun_out/r04_cfg5.log | cut -c1-200; echo "cfg5 $SECONDS s"
for i in 1 2 3; do python tools/host_path_bench.py 2>&1 | tail -2; done
echo "all $SECONDS s"
