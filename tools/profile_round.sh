#!/bin/bash
# tools/profile_round.sh <tag> [extra bench args]  -- run on the GPU box (via gpurun) from the repo root.
# 1) rocprofv3 --kernel-trace --stats of the default bench command
# 2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters), never combined with sys traces
# Summaries are written under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --cpu-sample 0 --no-extra "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -f csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra "$@" > /dev/null 2> $OUT/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS \
  -f csv -d $OUT/pmc_sq -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_tcc -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra "$@" > /dev/null 2> $OUT/pmc_tcc.err
# request-size resolved memory-side reads (gfx950 exposes 32/64/128-byte request counters; FETCH_SIZE tallies a
# 128-byte request as 64 bytes, MI355X_MICROARCH.md "HBM")
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -f csv -d $OUT/pmc_rdreq -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra "$@" > /dev/null 2> $OUT/pmc_rdreq.err
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum -f csv -d $OUT/pmc_dram -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra "$@" > /dev/null 2> $OUT/pmc_dram.err
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
