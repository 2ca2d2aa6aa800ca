set -u
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not 8gib" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | cut -c1-200
echo "tests: $SECONDS s"
bash tools/quick_bench.sh headline -- --steps 20 --warmup 5
bash tools/quick_bench.sh hit -- --steps 20 --warmup 5 --workload acgt_hit
bash tools/quick_bench.sh eng -- --steps 20 --warmup 5 --workload eng
bash tools/quick_bench.sh reads100 -- --steps 20 --warmup 5 --workload acgt_hit --plen 100 --npats 4000000
echo "all: $SECONDS s"
