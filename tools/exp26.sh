timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bench_two_ranks or comm_gather" --tb=long 2>&1 | tail -30 | cut -c1-300
