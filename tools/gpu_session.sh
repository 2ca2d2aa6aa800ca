set -u
SECONDS=0
timeout 300 bash tools/quick_bench.sh eng -- --steps 20 --warmup 5 --workload eng
timeout 300 bash tools/quick_bench.sh eng12 -- --steps 20 --warmup 5 --workload eng --len-range 12,12
timeout 300 bash tools/quick_bench.sh hit -- --steps 20 --warmup 5 --workload acgt_hit
timeout 300 bash tools/quick_bench.sh reads100 -- --steps 20 --warmup 5 --workload acgt_hit --plen 100 --npats 4000000
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not 8gib" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | cut -c1-200
echo "all: $SECONDS s"
