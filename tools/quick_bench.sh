#!/bin/bash
# tools/quick_bench.sh <label> [env assignments...] -- [bench args]: one compact result line
label=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
env "${envs[@]}" python bench.py --cpu-sample 20000 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$label', '|', round(d['value']/1e6,1), 'Mpat/s', round(d['ms_per_step'],2), 'ms/step frac', round(r['frac'],3), r['kernel'], 'count_ms', round(r['kernel_ms']-(r['locate_kernel_ms'] or 0),2), 'locate_ms', r['locate_kernel_ms'])"
