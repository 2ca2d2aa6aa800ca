source tools/exp_variants.sh exp13 true
timeout 600 python -m pytest tests/test_regexp.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
run hit_tm12 -- --workload acgt_hit
run hit_tm4 FEMTO_AMD_TAIL_MIN=4 -- --workload acgt_hit
run hit_tm3 FEMTO_AMD_TAIL_MIN=3 -- --workload acgt_hit
run eng_tm10 -- --workload eng
run eng_tm5 FEMTO_AMD_TAIL_MIN=5 -- --workload eng
run eng_tm3 FEMTO_AMD_TAIL_MIN=3 -- --workload eng
run reads100_tm4 FEMTO_AMD_TAIL_MIN=4 -- --workload acgt_hit --plen 100 --npats 4000000
python tools/host_path_bench.py 2>&1 | tail -1
