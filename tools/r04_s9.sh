set -u
SECONDS=0
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04/s9_tests.log 2>&1
grep -v "^RCCL\|^HIP \|^ROCm\|Hostname\|Librccl" gpurun_out/r04/s9_tests.log | tail -25 | cut -c1-300
echo "tests: $SECONDS s"
python - <<'PY'
import os, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from femto_amd import textgen as tg
from oracle import pyoracle as po
import femto_amd, numpy as np
path = "/tmp/femto_amd_bench/acgt_2p30_s20260928"
if not os.path.exists(path + "/_femto_index"):
    os.makedirs("/tmp/femto_amd_bench", exist_ok=True)
    femto_amd.build_index(path, [tg.t_acgt(1 << 30, 20260928)], params=None, infos=["bench"], device=0)
plen, flat = tg.p_rand(20, 10_000_000, 123)
po.write_fpat_flat("/tmp/p.fpat", plen, flat)
env = dict(os.environ, FEMTO_AMD_PIPE_TRACE="1")
for mode in ("count", "locate"):
    r = subprocess.run([po.REF_TOOL_AMD, "bench", path, "/tmp/p.fpat", mode, "100", "1", "8"], env=env, capture_output=True, text=True)
    print(mode, r.stdout.strip().splitlines()[-1][:300])
    print("\n".join(l for l in r.stderr.splitlines() if "femto_amd" in l)[:2000])
PY
echo "all: $SECONDS s"
