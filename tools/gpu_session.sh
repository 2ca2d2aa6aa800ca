set -u
SECONDS=0
timeout 70 bash tools/quick_bench.sh "headline 7w" -- --steps 20 --warmup 5
timeout 60 bash tools/quick_bench.sh "headline 8w" FEMTO_AMD_LIB=$PWD/ab/lib8.so -- --steps 20 --warmup 5
timeout 50 bash tools/quick_bench.sh "hit 7w" -- --steps 20 --warmup 5 --workload acgt_hit
timeout 50 bash tools/quick_bench.sh "hit 8w" FEMTO_AMD_LIB=$PWD/ab/lib8.so -- --steps 20 --warmup 5 --workload acgt_hit
echo "all: $SECONDS s"
