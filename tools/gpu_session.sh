set -u
SECONDS=0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -6 | cut -c1-220
echo "tests: $SECONDS s"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1100 bash tools/profile_round.sh r03_default --steps 20 --warmup 5 > gpurun_out/r03_default.log 2>&1; tail -3 gpurun_out/r03_default.log | cut -c1-200
echo "all: $SECONDS s"
