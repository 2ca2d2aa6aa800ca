timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "striped or bench_two_ranks or range_split_across" --tb=short 2>&1 | tail -15 | cut -c1-300
