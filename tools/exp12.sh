timeout 2400 python -m pytest tests -m gpu -x -q -k "not 8gib" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python tools/host_path_bench.py 2>&1 | tail -1
