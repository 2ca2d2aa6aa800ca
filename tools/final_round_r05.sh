# tools/final_round_r05.sh [phase...] -- on the GPU box (gpurun): the profile rounds profiles/r05_* are promoted from
# (python tools/promote_profiles.py r05_default ... afterwards, in the container).  Phases: default budget4x budget4x_hit
# mode0 mode1 regexp hit eng cfg5 ; no argument = all but cfg5.
set -u
SECONDS=0
B=4294967296
export TMPDIR=/tmp
PH="${*:-default budget4x budget4x_hit mode0 regexp hit eng}"
has() { case " $PH " in *" $1 "*) return 0;; esac; return 1; }
round() { # tag bench-args...
  t=$1; shift
  bash tools/profile_round.sh $t "$@" > gpurun_out/$t.log 2>&1; tail -2 gpurun_out/$t.log | cut -c1-200; echo "$t $SECONDS s"
}
has default && round r05_default --steps 20 --warmup 5
has budget4x && round r05_budget4x --steps 20 --warmup 5 --no-extra --open-opts hbm_budget_bytes=$B
has budget4x_hit && round r05_budget4x_hit --steps 10 --warmup 3 --no-extra --workload acgt_hit --open-opts hbm_budget_bytes=$B
has mode0 && FEMTO_AMD_RANK_MODE=raw round r05_mode0 --steps 2 --warmup 1 --no-extra --cpu-sample 0
has mode1 && FEMTO_AMD_RANK_MODE=lane round r05_mode1 --steps 3 --warmup 1 --no-extra
has hit && round r05_hit --steps 10 --warmup 3 --workload acgt_hit --no-extra
has eng && round r05_eng --steps 10 --warmup 3 --workload eng --no-extra
has cfg5 && round r05_cfg5 --steps 6 --warmup 2 --workload acgt_hit --text-log2 33 --no-extra --cpu-sample 20000 --ref-sample 10000
if has regexp; then
  # f4: the two automaton batches alone -- kernel trace, then separate counter passes (never combined with a trace domain)
  O=$PWD/gpurun_out/r05_regexp; mkdir -p $O
  python tools/regexp_bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o stats -- python tools/regexp_bench.py --reps 1 > $O/bench_stats.json 2> $O/stats.err
  for which in exact approx; do
    p() { n=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-include-regex "nfa_search_kernel" -f csv -d $O/pmc_${which}_$n -o pmc -- python tools/regexp_bench.py --reps 1 --which $which > /dev/null 2> $O/pmc_${which}_$n.err; }
    p fetch FETCH_SIZE
    p write WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
    p sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
    p occ SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  done
  python tools/summarize_regexp.py $O > $O/summary.txt 2>&1
  cat $O/summary.txt | cut -c1-220
  find $O -name "*.csv" -size +2M -delete
  echo "regexp $SECONDS s"
fi
echo "all $SECONDS s"
