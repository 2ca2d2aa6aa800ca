#!/usr/bin/env python3
"""tools/cfg3_histogram.py [--n N]: where the rank steps of BASELINE configs[2] go (round-4 verdict, task 1b).

cfg 3's count kernel starts every pattern with ONE table read -- the hashed 16-gram table (patterns of >= 16 symbols), the 9-gram
table (9..15), the level table (8) -- and then either compares the rest of the pattern with the text (range of ONE row: SA / text /
ISA, three requests) or takes rank steps (two requests each) until one row is left or the pattern ends.  This script measures,
with the engine's own count on SUFFIXES of the benchmark's patterns, per length class and per number of rows the table leaves:
how many patterns, and how many rank steps they take before the tail or the end.  Run on the GPU box after bench.py built the
sigma~96 index; prints a table (profiles/r05_cfg3_steps_histogram.txt)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--text-log2", type=int, default=30)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--tail-min", type=int, default=4)
    args = ap.parse_args()
    import torch  # noqa: F401
    import femto_amd
    from femto_amd import textgen as tg
    path = os.path.join(args.workdir, f"eng_2p{args.text_log2}_s{args.seed}")
    text = tg.t_eng_torch(1 << args.text_log2, args.seed, "cuda:0")
    if not os.path.exists(os.path.join(path, "_femto_index")):
        femto_amd.build_index(path, [text], params=None, infos=["bench"], device=0)
    ix = femto_amd.Index(path, device=0, options={"hbm_budget_bytes": femto_amd.BUDGET_ALL})
    pi = ix.pack_info()
    H1, H2, K = pi["context_syms"], pi["context2_syms"], pi["ktab_syms"]
    plen, flat = tg.p_hit(8, 64, args.n, args.seed + 3000, text)
    del text
    starts = tg.starts_of(plen)
    n = len(plen)
    T = np.where(plen >= H2, H2, np.where(plen >= H1, H1, np.minimum(plen, K))).astype(np.int32)     # symbols the first table read covers

    def count_suffix(idx, k):
        """rows of the last k[i] symbols of pattern idx[i]"""
        sl = k.astype(np.int32)
        st = (starts[idx] + plen[idx] - sl).astype(np.int64)
        f, l = ix.count_flat(sl, flat, st)
        return l - f + 1

    rows0 = count_suffix(np.arange(n), T)
    steps = np.zeros(n, dtype=np.int32)            # rank steps taken after the table
    tail = np.zeros(n, dtype=bool)                 # ended in the text tail
    k = T.copy()
    rows = rows0.copy()
    active = np.ones(n, dtype=bool)
    while active.any():
        # the kernel's rule (direct_kernels.hip.hpp): one row and at least tail_min symbols to go -> text tail; pattern done -> stop
        done = active & (k >= plen)
        active &= ~done
        t_ = active & (rows == 1) & (plen - k >= args.tail_min)
        tail |= t_
        active &= ~t_
        idx = np.flatnonzero(active)
        if not len(idx):
            break
        k[idx] += 1
        steps[idx] += 1
        rows[idx] = count_suffix(idx, k[idx])
    lc = np.digitize(plen, [9, 16, 32])            # 0: 8 | 1: 9-15 | 2: 16-31 | 3: 32-64
    rc = np.digitize(rows0, [2, 3, 5, 9, 101])     # 0: 1 | 1: 2 | 2: 3-4 | 3: 5-8 | 4: 9-100 | 5: > 100
    lnames = ["8", "9-15", "16-31", "32-64"]
    rnames = ["1", "2", "3-4", "5-8", "9-100", ">100"]
    print(f"cfg 3 rank steps after the first table read: {n} patterns sampled like bench.py's batch (lengths 8..64), tables: level K={K}, context H={H1}, wide H2={H2}; tail_min {args.tail_min}")
    print(f"all patterns: {steps.sum() / n:.3f} rank steps per pattern, {tail.mean() * 100:.1f} % end in the text tail, {(steps == 0).mean() * 100:.1f} % take no rank step at all")
    print("%-7s %-7s %9s %9s %11s %10s" % ("length", "rows", "patterns%", "steps/pat", "share steps%", "tail%"))
    tot = max(1, int(steps.sum()))
    for a in range(4):
        for b in range(6):
            m = (lc == a) & (rc == b)
            if not m.any():
                continue
            print("%-7s %-7s %9.2f %9.2f %11.2f %10.1f" % (lnames[a], rnames[b], 100 * m.mean(), steps[m].mean(), 100 * steps[m].sum() / tot, 100 * tail[m].mean()))
    for b in range(6):
        m = rc == b
        if m.any():
            print("%-7s %-7s %9.2f %9.2f %11.2f %10.1f" % ("all", rnames[b], 100 * m.mean(), steps[m].mean(), 100 * steps[m].sum() / tot, 100 * tail[m].mean()))
    # what a parallel multi-row tail could save: patterns of >= H2 symbols that leave the wide table with 2..4 rows
    m = (plen >= H2) & (rows0 >= 2) & (rows0 <= 4)
    print(f"patterns of >= {H2} symbols leaving the wide table with 2..4 rows: {100 * m.mean():.2f} % of the batch, {100 * steps[m].sum() / tot:.1f} % of all rank steps "
          f"({steps[m].mean() if m.any() else 0:.2f} steps each = {2 * steps[m].mean() if m.any() else 0:.1f} requests before a tail of 3)")
    ix.close()


if __name__ == "__main__":
    main()
