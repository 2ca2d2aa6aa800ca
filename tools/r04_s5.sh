set -u
SECONDS=0
mkdir -p gpurun_out/r04
A="--text-log2 33 --workload acgt_hit --steps 6 --warmup 2 --pmc off --cpu-sample 0"
timeout 900 bash tools/quick_bench.sh "cfg5 ru" -- $A
echo "first: $SECONDS s"
timeout 400 bash tools/quick_bench.sh "cfg5 noru" -- $A --open-opts rank_units=0
timeout 400 bash tools/quick_bench.sh "cfg5 ru" -- $A
timeout 400 bash tools/quick_bench.sh "cfg5 noru" -- $A --open-opts rank_units=0
timeout 400 bash tools/quick_bench.sh "cfg5 ru rand" -- --text-log2 33 --steps 6 --warmup 2 --pmc off --cpu-sample 0
timeout 400 bash tools/quick_bench.sh "cfg5 noru rand" -- --text-log2 33 --steps 6 --warmup 2 --pmc off --cpu-sample 0 --open-opts rank_units=0
echo "all: $SECONDS s"
