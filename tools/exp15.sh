# diagnose the random-index failure, then the whole suite without -x, then fuzz + soak
T='tests/test_gpu_parity.py::test_random_indexes_vs_oracle[25]'
timeout 300 python -m pytest "$T" -q --tb=long > gpurun_out/exp15_seed25.log 2>&1; tail -1 gpurun_out/exp15_seed25.log
FEMTO_AMD_TAIL_MIN=12 timeout 300 python -m pytest "$T" -q --tb=line > gpurun_out/exp15_seed25_tm12.log 2>&1; echo "tail_min=12: $(tail -1 gpurun_out/exp15_seed25_tm12.log)"
FEMTO_AMD_DENSE=0 timeout 300 python -m pytest "$T" -q --tb=line > gpurun_out/exp15_seed25_nodense.log 2>&1; echo "dense=0: $(tail -1 gpurun_out/exp15_seed25_nodense.log)"
FEMTO_AMD_IND=0 timeout 300 python -m pytest "$T" -q --tb=line > gpurun_out/exp15_seed25_noind.log 2>&1; echo "ind=0: $(tail -1 gpurun_out/exp15_seed25_noind.log)"
FEMTO_AMD_KTAB=0 timeout 300 python -m pytest "$T" -q --tb=line > gpurun_out/exp15_seed25_noktab.log 2>&1; echo "ktab=0: $(tail -1 gpurun_out/exp15_seed25_noktab.log)"
timeout 2700 python -m pytest tests -m gpu -q --tb=short > gpurun_out/exp15_suite.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/exp15_suite.log | tail -15 | cut -c1-250
timeout 900 python tools/fuzz_gpu.py 1 140 > gpurun_out/exp15_fuzz.log 2>&1; tail -2 gpurun_out/exp15_fuzz.log | cut -c1-250
FEMTO_AMD_SOAK_ROWS=300000,3000000 timeout 900 python tools/soak.py 12 > gpurun_out/exp15_soak.log 2>&1; tail -2 gpurun_out/exp15_soak.log | cut -c1-250
