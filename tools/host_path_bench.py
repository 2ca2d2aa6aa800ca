#!/usr/bin/env python3
"""tools/host_path_bench.py: the reference's own calling convention (pageable host arrays in and out) on the bench index:
femto_amd_count_flat of 10 M random 20-mers, best of 5 after a warm-up call, for the staging knobs in the environment."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import femto_amd  # noqa: E402
from femto_amd import textgen as tg  # noqa: E402

path = os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench") + "/acgt_2p30_s20260928"
n = int(os.environ.get("NPATS", "10000000"))
ix = femto_amd.Index(path, device=0)
plen, flat = tg.p_rand(20, n, 123)
starts = tg.starts_of(plen)
first = np.ones(n, dtype=np.int64)
last = np.ones(n, dtype=np.int64)
L = femto_amd.lib()
best = 1e9
for rep in range(6):
    t0 = time.perf_counter()
    rc = L.femto_amd_count_flat(ix.handle, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data, first.ctypes.data, last.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    if rep:
        best = min(best, dt)
print("host-pointer count: %.2f ms  %.2f G patterns/s  (threads %s chunk %s keys %s)" % (
    1e3 * best, n / best / 1e9, os.environ.get("FEMTO_AMD_HOST_THREADS", "default"), os.environ.get("FEMTO_AMD_PIPE_CHUNK_LOG2", "21"),
    os.environ.get("FEMTO_AMD_HOST_KEYS", "1")), flush=True)
