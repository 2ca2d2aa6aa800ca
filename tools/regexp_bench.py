#!/usr/bin/env python3
"""tools/regexp_bench.py [--which exact|approx|both] [--n N] [--text-log2 K]: the two automaton batches of bench.py's `regexp_batch`
extra (benchlib/extras.py regexp_workloads) on the bench index, alone -- the command rocprofv3 profiles for
profiles/r05_regexp_stats.txt (`--kernel-trace --stats`, and separate `--pmc` passes).  Prints one JSON line per batch:
wall ms of the femto_amd_nfa_search_batch call, the nfa_search_kernel's own time (HIP events), automata/s."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="both", choices=["exact", "approx", "both"])
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--text-log2", type=int, default=30)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--workdir", default=os.environ.get("FEMTO_AMD_BENCH_DIR", "/tmp/femto_amd_bench"))
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch  # noqa: F401
    import femto_amd
    from femto_amd import textgen as tg
    from benchlib.extras import regexp_workloads
    path = os.path.join(args.workdir, f"acgt_2p{args.text_log2}_s{args.seed}")
    if not os.path.exists(os.path.join(path, "_femto_index")):
        os.makedirs(args.workdir, exist_ok=True)
        femto_amd.build_index(path, [tg.t_acgt(1 << args.text_log2, args.seed)], params=None, infos=["bench"], device=0)
    ix = femto_amd.Index(path, device=0, options={"hbm_budget_bytes": femto_amd.BUDGET_ALL})
    work = regexp_workloads(femto_amd, args.seed, args.n, args.n)
    ix.nfa_search_batch(work["exact_motifs_14_18"][:128], max_results=1 << 22)
    for name, nfas in work.items():
        if args.which != "both" and not name.startswith(args.which):
            continue
        pre = femto_amd.NfaBatch(nfas)
        for rep in range(args.reps):
            ix.kernel_time_reset()
            ix.kernel_time_enable(True)
            t0 = time.perf_counter()
            r = ix.nfa_search_batch(pre, max_results=1 << 25)
            dt = time.perf_counter() - t0
            ix.kernel_time_enable(False)
            k_ms, k_n = ix.kernel_time("regexp")
            print(json.dumps({"batch": name, "rep": rep, "automata": len(nfas), "nodes_avg": float(np.mean([a.num_nodes for a in nfas])),
                              "wall_ms": 1e3 * dt, "automata_per_s": len(nfas) / dt, "kernel_ms_total": k_ms * k_n, "kernel_launches": k_n,
                              "kernel_automata_per_s": len(nfas) / (k_ms * k_n * 1e-3) if k_n else None, "result_ranges": int(len(r[1])),
                              "not_ok": int((r[5] != 0).sum())}), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
