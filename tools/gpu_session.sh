set -u
mkdir -p gpurun_out/s14
bash tools/quick_bench.sh headline -- --pmc off --steps 20 --warmup 5 2>&1 | tee gpurun_out/s14/bench.txt
bash tools/quick_bench.sh hit -- --pmc off --workload acgt_hit --steps 10 --warmup 3 2>&1 | tee -a gpurun_out/s14/bench.txt
bash tools/quick_bench.sh reads100 -- --pmc off --workload acgt_hit --plen 100 --npats 4000000 --steps 10 --warmup 3 2>&1 | tee -a gpurun_out/s14/bench.txt
bash tools/quick_bench.sh eng_default -- --pmc off --workload eng --steps 10 --warmup 3 2>&1 | tee -a gpurun_out/s14/bench.txt
