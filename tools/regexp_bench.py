#!/usr/bin/env python3
"""tools/regexp_bench.py: batched automaton search (femto_amd_nfa_search_batch = do_regexp_query for many automata, one
workgroup each) on the bench index: N random DNA motifs with classes, alternations and optional symbols, exact and APPROX 1;
GPU timing only -- bench.py's `extra.regexp_batch` line times the genuine reference beside a batch and compares the result lists."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import femto_amd  # noqa: E402

path = os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench") + "/acgt_2p30_s20260928"
N = int(os.environ.get("NREGEX", "20000"))
rng = np.random.Generator(np.random.PCG64(7))


def motif(k):
    out = b""
    for _ in range(k):
        r = rng.random()
        if r < 0.70:
            out += bytes([b"ACGT"[rng.integers(0, 4)]])
        elif r < 0.85:
            out += b"[" + bytes(sorted(set(b"ACGT"[i] for i in rng.integers(0, 4, 2)))) + b"]"
        elif r < 0.93:
            out += b"(" + bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2)) + b"|" + bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2)) + b")"
        else:
            out += bytes([b"ACGT"[rng.integers(0, 4)]]) + b"?"
    return out


ix = femto_amd.Index(path, device=0)
for what, approx, k in (("exact motifs of 14-18 terms", None, (14, 19)), ("APPROX 1 motifs of 16-20 terms", (1, 1, 1, 1), (16, 21))):
    pats = [motif(int(rng.integers(*k))) for _ in range(N)]
    nfas = [femto_amd.Nfa.from_regex(p, approx) for p in pats]
    ix.nfa_search_batch(nfas[:256], max_results=1 << 22)          # warm-up (scratch, arena)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        start, first, last, mlen, cost, status = ix.nfa_search_batch(nfas, max_results=1 << 24)
        best = min(best, time.perf_counter() - t0)
    nodes = np.mean([a.num_nodes for a in nfas])
    line = "%-34s %6d automata (%.0f nodes avg): %.1f ms per batch = %.0f automata/s, %d result ranges, %d not ok" % (
        what, N, nodes, 1e3 * best, N / best, len(first), int((status != 0).sum()))
    print(line, flush=True)
ix.close()
