#!/usr/bin/env python3
"""tools/summarize_regexp.py <dir>: the f4 profile round of tools/final_round_r05.sh -> a short text report: the batch lines of
tools/regexp_bench.py, rocprofv3's kernel stats of the same command, and per-batch means of the separate --pmc passes for
nfa_search_kernel (memory-side bytes and requests per launch, wave cycles, LDS use)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]

print("== tools/regexp_bench.py (20 000 automata per femto_amd_nfa_search_batch call, 1 GiB ACGT index)")
for name in ("bench.json", "bench_stats.json"):
    try:
        print("-- " + name)
        print(open(os.path.join(out, name)).read().strip())
    except OSError as e:
        print("missing", e)

print("\n== rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
for f in sorted(glob.glob(os.path.join(out, "stats/**/*kernel_stats.csv"), recursive=True)):
    with open(f) as fh:
        for r in list(csv.DictReader(fh))[:8]:
            print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

print("\n== PMC, nfa_search_kernel, per batch (sum over the batch's launches; n launches)")
for which in ("exact", "approx"):
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out, f"pmc_{which}_*/**/*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "nfa_search_kernel" in r.get("Kernel_Name", ""):
                    acc[r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "0")))
    print(which)
    for c, v in sorted(acc.items()):
        print(f"   {c:28s} sum {sum(v):.6g}  mean {sum(v) / len(v):.6g}  n={len(v)}")
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        b = 2 * sum(acc["FETCH_SIZE"]) * 1024 + sum(acc["WRITE_SIZE"]) * 1024
        print(f"   memory-side bytes (2 x FETCH_SIZE KiB + WRITE_SIZE KiB, incl. Infinity Cache hits): {b / 1e9:.3f} GB per batch")
