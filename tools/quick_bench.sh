#!/bin/bash
# tools/quick_bench.sh <label> [env assignments...] -- [bench args]: one compact result line
label=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
env "${envs[@]}" python bench.py --cpu-sample 20000 --no-extra "$@" 2>/tmp/qb.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d.get('roofline') or {}
    print('$label', '|', round(d['value']/1e6,1), 'Mpat/s', round(d['ms_per_step'],2), 'ms/step frac', round(r.get('frac') or 0,3), r.get('kernel'), 'count_ms', round(r.get('count_kernel_ms') or 0,2), 'locate_ms', round(r.get('locate_kernel_ms') or 0,2), 'rows', d['config']['located_rows_per_gpu'], 'K', d['config']['index']['structures']['level_table_syms'], 'marks', d['config']['index']['structures']['mark_every'], 'ru', d['config']['index']['structures']['rank_units']>>20, 'MB hbm', d['config']['index']['structures']['hbm_allocated']>>20, 'MB')
except Exception as e:
    print('$label FAILED', e, t[:300]); print(open('/tmp/qb.err').read()[-1500:])"
