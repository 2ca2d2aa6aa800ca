#!/usr/bin/env python3
"""tools/striped_bench.py: what the striped layout's address translation costs on ONE GPU -- the bench index opened plainly
and as a striped handle whose stripes all live on GPU 0 (same kernels, same arrays; the only difference is that the big
arrays are HIP virtual-memory mappings), 10 M random 20-mers through the enqueue-only chain."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import femto_amd  # noqa: E402
from femto_amd import textgen as tg  # noqa: E402

lg = int(os.environ.get("TEXT_LOG2", "30"))      # 33: BASELINE configs[4]'s 8 GiB text (is every line a TLB miss because of HOW the arrays are mapped?)
path = os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench") + f"/acgt_2p{lg}_s20260928"
n = int(os.environ.get("NPATS", "10000000"))
dev = torch.device("cuda", 0)
if os.environ.get("WORKLOAD", "rand") == "hit":
    plen, flat = tg.p_hit(20, 20, n, 123, np.asarray(np.load(path + ".text.npy", mmap_mode="r")))
else:
    plen, flat = tg.p_rand(20, n, 123)
st = torch.cuda.current_stream().cuda_stream
for what in ("plain", "striped x2 on one GPU"):
    if what == "plain":
        keep, ix = None, femto_amd.Index(path, device=0)
    else:
        keep = femto_amd.Index(path, devices=[0, 0], striped=True)
        ix = keep.child(0)
    b = bench.Batch(torch, dev, plen, flat)
    b.settle(ix, 100, st)
    for _ in range(5):
        b.step(ix, 100, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        b.step(ix, 100, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-24s %.3f ms/step  %.2f G patterns/s  (K = %d)" % (what, 1e3 * dt, n / dt / 1e9, ix.pack_info()["ktab_syms"]), flush=True)
    del b
    ix.close()
    if keep is not None:
        keep.close()
