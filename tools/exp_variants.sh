#!/bin/bash
# tools/exp_variants.sh <tag> -- run on the GPU box: bench lines of the pipeline variants (no CPU baseline, no extras)
TAG=${1:-exp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 20 --warmup 5 --no-extra --cpu-sample 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(f"{sys.argv[2]:28s} {d['value']/1e9:7.3f} G/s  {d['ms_per_step']:7.3f} ms/step  rows {d['config']['located_rows_per_gpu']}  ktab {d['config']['index']['packed_lines']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
"$@"
