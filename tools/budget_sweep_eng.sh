#!/bin/bash
# tools/budget_sweep_eng.sh -- on the GPU box: what a GB of HBM buys on a BYTE alphabet (cfg 3: 1 GiB sigma~96 text, 10 M sampled patterns of 8..64
# symbols, count + locate max_occs 100): the library's default bound (8 x text), 16 / 32 / 64 x text, everything -- each with rows and row-free
for b in -1 16 32 64; do
  if [ "$b" = "-1" ]; then B=-1; L="default (8x)"; else B=$(python3 -c "print(int($b * (1 << 30)))"); L="${b}x"; fi
  timeout 400 bash tools/quick_bench.sh "eng budget $L" -- --steps 20 --warmup 3 --pmc off --workload eng --open-opts hbm_budget_bytes=$B
  timeout 400 bash tools/quick_bench.sh "eng budget $L row-free" -- --steps 20 --warmup 3 --pmc off --workload eng --row-free --open-opts hbm_budget_bytes=$B
done
timeout 400 bash tools/quick_bench.sh "eng everything" -- --steps 20 --warmup 3 --pmc off --workload eng
timeout 400 bash tools/quick_bench.sh "eng everything row-free" -- --steps 20 --warmup 3 --pmc off --workload eng --row-free
