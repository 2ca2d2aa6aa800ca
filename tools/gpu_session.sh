set -u
mkdir -p gpurun_out/s6
bash tools/quick_bench.sh headline -- --pmc off --steps 20 --warmup 5 2>&1 | tee gpurun_out/s6/bench.txt
bash tools/quick_bench.sh hit -- --pmc off --workload acgt_hit --steps 10 --warmup 3 2>&1 | tee -a gpurun_out/s6/bench.txt
bash tools/quick_bench.sh reads100 -- --pmc off --workload acgt_hit --plen 100 --npats 4000000 --steps 10 --warmup 3 2>&1 | tee -a gpurun_out/s6/bench.txt
bash tools/quick_bench.sh eng_ctx2_16 FEMTO_AMD_CTX2_SYMS=16 FEMTO_AMD_CTX2_MB=90000 -- --pmc off --workload eng --steps 6 --warmup 2 2>&1 | tee -a gpurun_out/s6/bench.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -15 | tee gpurun_out/s6/parity.txt
