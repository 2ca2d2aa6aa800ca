#!/usr/bin/env python3
"""tools/host_locate_bench.py: parallel_locate's calling convention on the bench index -- pageable host arrays in, one
malloc()ed offsets array out (femto_amd_locate_flat_alloc): 10 M 20-mers sampled from the text (every pattern located)
and 10 M random ones, best of 4 after a warm-up call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import femto_amd  # noqa: E402
from femto_amd import textgen as tg  # noqa: E402

base = os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench") + "/acgt_2p30_s20260928"
n = int(os.environ.get("NPATS", "10000000"))
ix = femto_amd.Index(base, device=0)
text = np.load(base + ".text.npy", mmap_mode="r")
for what, (plen, flat) in (("sampled (all located)", tg.p_hit(20, 20, n, 5, np.asarray(text))), ("random", tg.p_rand(20, n, 123))):
    starts = tg.starts_of(plen)
    best, rows = 1e9, 0
    import ctypes as C
    L = femto_amd.lib()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    noccs = np.ones(n, dtype=np.int32)            # touched once: the C caller's own array
    for rep in range(5):
        total, buf = C.c_int64(0), C.c_void_p()
        t0 = time.perf_counter()
        rc = L.femto_amd_locate_flat_alloc(ix.handle, n, plen.ctypes.data, flat.ctypes.data, starts.ctypes.data, 100, noccs.ctypes.data, None,
                                           C.byref(buf), C.byref(total))
        dt = time.perf_counter() - t0
        assert rc == 0
        rows = total.value
        libc.free(buf)
        if rep:
            best = min(best, dt)
    print("host-pointer locate, %-22s %.2f ms  %.2f G patterns/s  (%d rows)" % (what, 1e3 * best, n / best / 1e9, rows), flush=True)
