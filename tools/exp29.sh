timeout 900 python tools/fuzz_gpu.py 6 140 > gpurun_out/exp29_fuzz6.log 2>&1; tail -1 gpurun_out/exp29_fuzz6.log | cut -c1-300
( time python bench.py > gpurun_out/exp29_default_bench.json 2> gpurun_out/exp29_default_bench.err ) 2>&1 | grep real
bash tools/exp24.sh
