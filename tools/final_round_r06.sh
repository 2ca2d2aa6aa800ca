# tools/final_round_r06.sh [phase...] -- on the GPU box (gpurun): the profile rounds profiles/r06_* are promoted from
# (python tools/promote_profiles.py r06_default ... afterwards, in the container).  Every round times 200 steps (round-5 verdict:
# a 20-step window is two orders below scheduler noise).  Phases: default hit hit_rowfree eng eng_rowfree eng_default
# eng_default_rowfree budget4x budget4x_hit mode0 mode1 regexp cfg5 ; no argument = all but cfg5 and mode1.
set -u
SECONDS=0
B=4294967296
export TMPDIR=/tmp
PH="${*:-default hit hit_rowfree eng eng_rowfree eng_default eng_default_rowfree budget4x budget4x_hit mode0 regexp}"
has() { case " $PH " in *" $1 "*) return 0;; esac; return 1; }
round() { # tag bench-args...
  t=$1; shift
  bash tools/profile_round.sh $t "$@" > gpurun_out/$t.log 2>&1; tail -2 gpurun_out/$t.log | cut -c1-200; echo "$t $SECONDS s"
}
has default && round r06_default --steps 200 --warmup 10
has hit && round r06_hit --steps 200 --warmup 10 --workload acgt_hit --no-extra
has hit_rowfree && round r06_hit_rowfree --steps 200 --warmup 10 --workload acgt_hit --row-free --no-extra
has eng && round r06_eng --steps 200 --warmup 10 --workload eng --no-extra
has eng_rowfree && round r06_eng_rowfree --steps 200 --warmup 10 --workload eng --row-free --no-extra
has eng_default && round r06_eng_default --steps 50 --warmup 5 --workload eng --no-extra --open-opts hbm_budget_bytes=-1
has eng_default_rowfree && round r06_eng_default_rowfree --steps 50 --warmup 5 --workload eng --row-free --no-extra --open-opts hbm_budget_bytes=-1
has budget4x && round r06_budget4x --steps 200 --warmup 10 --no-extra --open-opts hbm_budget_bytes=$B
has budget4x_hit && round r06_budget4x_hit --steps 100 --warmup 5 --no-extra --workload acgt_hit --open-opts hbm_budget_bytes=$B
has mode0 && FEMTO_AMD_RANK_MODE=raw round r06_mode0 --steps 2 --warmup 1 --no-extra --cpu-sample 0
has mode1 && FEMTO_AMD_RANK_MODE=lane round r06_mode1 --steps 5 --warmup 1 --no-extra
has cfg5 && round r06_cfg5 --steps 50 --warmup 5 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 20000 --ref-sample 10000
if has regexp; then
  # f4: the two automaton batches alone (kernel trace, then separate counter passes -- never combined with a trace domain), then
  # concurrent callers on one handle
  O=$PWD/gpurun_out/r06_regexp; mkdir -p $O
  FEMTO_AMD_NFA_STATS=1 python tools/regexp_bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o stats -- python tools/regexp_bench.py --reps 1 > $O/bench_stats.json 2> $O/stats.err
  for which in exact approx; do
    p() { n=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-include-regex "nfa_search_kernel" -f csv -d $O/pmc_${which}_$n -o pmc -- python tools/regexp_bench.py --reps 1 --which $which > /dev/null 2> $O/pmc_${which}_$n.err; }
    p fetch FETCH_SIZE
    p write WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
    p sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
  done
  python tools/summarize_regexp.py $O > $O/summary.txt 2>&1
  cat $O/summary.txt | cut -c1-220
  for T in 4 8; do
    python tools/regexp_bench.py --which approx --reps 2 --concurrent $T 2>/dev/null | grep -v nodes_avg > $O/concurrent_$T.json
    GPU_MAX_HW_QUEUES=4 python tools/regexp_bench.py --which approx --reps 1 --concurrent $T 2>/dev/null | grep concurrent > $O/concurrent_${T}_4queues.json
    FEMTO_AMD_NFA_FAIR=0 python tools/regexp_bench.py --which approx --reps 1 --concurrent $T 2>/dev/null | grep concurrent > $O/concurrent_${T}_nofair.json
  done
  find $O -name "*.csv" -size +2M -delete
  echo "regexp $SECONDS s"
fi
echo "all $SECONDS s"
