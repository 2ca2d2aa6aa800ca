#!/bin/bash
# tools/budget_sweep.sh -- on the GPU box: what a GB of HBM buys.  The headline batch (10 M random 20-mers) and sampled 20-mers (with rows / row-free)
# on the 1 GiB DNA index opened with hbm_budget_bytes = 2.5 ... 32 x the text (femto_amd_open_opts), then without a budget.
for gb in 2.5 3 4 8 16 32; do
  B=$(python3 -c "print(int($gb * (1 << 30)))")
  timeout 120 bash tools/quick_bench.sh "budget ${gb}x rand" -- --steps 50 --warmup 5 --pmc off --open-opts hbm_budget_bytes=$B
  timeout 120 bash tools/quick_bench.sh "budget ${gb}x hit " -- --steps 20 --warmup 3 --pmc off --workload acgt_hit --open-opts hbm_budget_bytes=$B
  timeout 120 bash tools/quick_bench.sh "budget ${gb}x hit row-free" -- --steps 20 --warmup 3 --pmc off --workload acgt_hit --row-free --open-opts hbm_budget_bytes=$B
done
timeout 120 bash tools/quick_bench.sh "no budget rand" -- --steps 50 --warmup 5 --pmc off
timeout 120 bash tools/quick_bench.sh "no budget hit " -- --steps 20 --warmup 3 --pmc off --workload acgt_hit
timeout 120 bash tools/quick_bench.sh "no budget hit row-free" -- --steps 20 --warmup 3 --pmc off --workload acgt_hit --row-free
