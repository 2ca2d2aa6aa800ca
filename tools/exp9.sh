source tools/exp_variants.sh exp9 true
timeout 1800 python -m pytest tests/test_integration.py tests/test_gpu_parity.py -m gpu -x -q -k "not 8gib" 2>&1 | grep -E "passed|failed|Error|error" | tail -6
run acgt --
run hit -- --workload acgt_hit
run eng -- --workload eng
run reads100 -- --workload acgt_hit --plen 100 --npats 4000000
