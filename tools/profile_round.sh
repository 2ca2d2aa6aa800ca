#!/bin/bash
# tools/profile_round.sh <tag> [extra bench args]  -- run on the GPU box (via gpurun) from the repo root.
# 1) the bench line itself (with its own live PMC traffic pass), 2) rocprofv3 --kernel-trace --stats of the same command,
# 3) separate --pmc passes (memory-side request counters, SQ counters, L2 hit/miss, address translation), never combined with
#    any trace domain.  The passes profile the bench process itself (--pmc off inside), so they also work for an index of which
#    the GPU cannot hold a second copy (cfg 5).
# Summaries are written under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r02}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RX="count_direct_kernel|locate_walk_kernel|count_kernel|locate_kernel|count_tail_kernel|plan_rows_kernel|plan_super_kernel|count_keys_kernel"
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- python bench.py --cpu-sample 0 --no-extra --pmc off "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
pass() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "$RX" -f csv -d $OUT/pmc_$n -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra --pmc off "${ARGS[@]}" > /dev/null 2> $OUT/pmc_$n.err
}
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
# address translation: UTCL1 (per-CU TLB) requests / hits / misses and the time the UTCL2 is busy (cfg 5: is every line a TLB miss?)
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY
# the CU's vector-memory address path: per-lane cache accesses and the requests that leave the L1 (the
# count kernel of a byte alphabet is bound by accesses per CU cycle, not by bytes)
pass ta SQ_WAVES SQ_INSTS_VMEM_RD TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-260
find $OUT -name "*.csv" -size +2M -delete
