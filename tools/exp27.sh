timeout 900 python tools/fuzz_gpu.py 4 140 > gpurun_out/exp27_fuzz4.log 2>&1; tail -2 gpurun_out/exp27_fuzz4.log | cut -c1-400
timeout 900 python tools/fuzz_gpu.py 5 140 > gpurun_out/exp27_fuzz5.log 2>&1; tail -2 gpurun_out/exp27_fuzz5.log | cut -c1-400
FEMTO_AMD_SOAK_ROWS=300000,3000000 timeout 1200 python tools/soak.py 16 > gpurun_out/exp27_soak.log 2>&1; tail -2 gpurun_out/exp27_soak.log | cut -c1-250
bash tools/exp24.sh
