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
    ap.add_argument("--concurrent", type=int, default=0, help="this many caller threads, each with an APPROX 1 batch of its own, on ONE handle at once "
                                                               "(the reference keeps many do_regexp_query state machines in flight, server.c:3969-4001)")
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
    if args.concurrent > 0:
        concurrent(args, ix, femto_amd)
    ix.close()


def concurrent(args, ix, femto_amd):
    """`--concurrent T`: T caller threads, each with its own batch of --n APPROX 1 motifs (seeds of their own), calling
    femto_amd_nfa_search_batch on one handle at the same time.  Every thread's result lists are compared with the lists the same
    batch returned when it ran alone; prints the wall time of all T calls and the workgroup occupancy over that time."""
    import threading
    from benchlib.extras import regexp_workloads
    T = args.concurrent
    batches = [femto_amd.NfaBatch(regexp_workloads(femto_amd, args.seed + 1000 * (t + 1), 1, args.n)["approx1_motifs_16_20"]) for t in range(T)]
    alone, slots = [], None
    for b in batches:      # each batch alone: the answers, and the shader clock (cycles of the span / the kernel's milliseconds)
        ix.kernel_time_reset()
        ix.kernel_time_enable(True)
        t0 = time.perf_counter()
        r = ix.nfa_search_batch(b, max_results=1 << 25)
        dt = time.perf_counter() - t0
        ix.kernel_time_enable(False)
        k_ms, k_n = ix.kernel_time("regexp")
        st = ix.nfa_stats()
        alone.append((r, dt, st))
        slots = st["workgroups"]
        print(json.dumps({"batch": "approx1 alone", "automata": b.n, "wall_ms": 1e3 * dt, "kernel_ms": k_ms * k_n, "automata_per_s": b.n / dt, "stats": st}), flush=True)
    for rep in range(args.reps):
        out, stats = [None] * T, [None] * T
        go = threading.Barrier(T + 1)

        def run(t):
            go.wait()
            out[t] = ix.nfa_search_batch(batches[t], max_results=1 << 25)
            stats[t] = ix.nfa_stats(thread=True)
        th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        go.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        same = all(all(np.array_equal(a, b) for a, b in zip(out[t], alone[t][0])) for t in range(T))
        assert same, "concurrent callers: result lists differ from the lists of the same batches run alone"
        busy = sum(s["busy_s"] for s in stats)
        occ = busy / (slots * dt) if slots else None      # busy workgroup-seconds of all T batches over (slots of the GPU x wall time of the T calls)
        print(json.dumps({"batch": f"approx1 x {T} concurrent callers", "rep": rep, "automata": sum(b.n for b in batches), "wall_ms": 1e3 * dt,
                          "automata_per_s": sum(b.n for b in batches) / dt, "sum_of_alone_ms": 1e3 * sum(a[1] for a in alone),
                          "longest_alone_ms": 1e3 * max(a[1] for a in alone), "workgroup_occupancy": occ, "workgroup_slots": slots,
                          "per_call_search_span_ms": [1e3 * s["span_s"] for s in stats], "longest_pops": [s["pops_longest"] for s in stats],
                          "longest_waited_ms": [1e3 * s["longest_waited_s"] for s in stats],
                          "equal_to_alone": same}), flush=True)


if __name__ == "__main__":
    main()
