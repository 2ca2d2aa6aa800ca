#!/bin/bash
# tools/ab_bench.sh <other libfemto_amd.so> [bench args] -- on the GPU box: the in-tree library and another build of it (an
# older commit, another tuning constant) on the SAME box, alternating, so that box-to-box spread (+-4 % on this pool) does not
# decide an A/B.  Build the other library with FEMTO_AMD_LIB=<path> python -c "import femto_amd.build as b; b.build()" in a
# checkout of the other commit and put it somewhere inside the repo (it travels with the gpurun snapshot).
OTHER=$1; shift
for i in 1 2; do
  bash tools/quick_bench.sh "in-tree" -- --steps 20 --warmup 5 "$@"
  bash tools/quick_bench.sh "other  " FEMTO_AMD_LIB=$OTHER -- --steps 20 --warmup 5 "$@"
done
