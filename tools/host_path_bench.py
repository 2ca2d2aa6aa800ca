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
import ctypes as C
L.femto_amd_host_pipeline_stats.argtypes = [C.c_void_p, C.c_void_p]
best, best_stats = 1e9, None
times = []
for rep in range(int(os.environ.get("REPS", "8"))):
    t0 = time.perf_counter()
    rc = L.femto_amd_count_flat(ix.handle, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data, first.ctypes.data, last.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    st = np.zeros(8)
    L.femto_amd_host_pipeline_stats(ix.handle, st.ctypes.data)
    if rep:
        times.append(dt)
    if rep and dt < best:
        best, best_stats = dt, st
names = ("stage_ms", "wait_in_ms", "enqueue_ms", "wait_out_ms", "unpack_ms", "call_ms", "chunks", "threads")
print("  breakdown of the best call:", {k: round(float(v), 3) for k, v in zip(names, best_stats)}, flush=True)
print("  all calls after the warm-up (ms):", " ".join("%.1f" % (1e3 * t) for t in times), " mean %.2f" % (1e3 * sum(times) / len(times)), flush=True)
print("host-pointer count: %.2f ms  %.2f G patterns/s  (threads %s chunk %s keys %s)" % (
    1e3 * best, n / best / 1e9, os.environ.get("FEMTO_AMD_HOST_THREADS", "default"), os.environ.get("FEMTO_AMD_PIPE_CHUNK_LOG2", "21"),
    os.environ.get("FEMTO_AMD_HOST_KEYS", "1")), flush=True)
