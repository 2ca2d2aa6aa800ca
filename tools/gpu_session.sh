set -u
SECONDS=0
timeout 520 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | cut -c1-220
echo "tests: $SECONDS s"
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 200 bash tools/quick_bench.sh headline -- --steps 20 --warmup 5
echo "all: $SECONDS s"
