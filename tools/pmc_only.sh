#!/bin/bash
# tools/pmc_only.sh <tag> <counter list...> -- [bench args]: one extra --pmc pass, summarised
TAG=$1; shift
ctr=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ctr+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc "${ctr[@]}" -f csv -d $OUT/pmc_x -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extra --pmc off "$@" > /dev/null 2> $OUT/pmc_x.err
python tools/summarize_profile.py $OUT | tail -20
