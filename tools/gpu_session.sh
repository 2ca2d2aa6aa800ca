set -u
SECONDS=0
export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0,'.')
import femto_amd
from femto_amd import textgen as tg
import os
p='/tmp/femto_amd_bench/acgt_2p30_s20260928'
if not os.path.exists(p+'/_femto_index'):
    os.makedirs('/tmp/femto_amd_bench', exist_ok=True)
    femto_amd.build_index(p, [tg.t_acgt(1<<30, 20260928)], params=None, infos=['bench'], device=0)
"
for i in 1 2; do
  for q in 4 16; do
    echo "GPU_MAX_HW_QUEUES=$q"
    GPU_MAX_HW_QUEUES=$q python tools/host_path_bench.py 2>/dev/null | tail -2
  done
done
echo "all $SECONDS s"
