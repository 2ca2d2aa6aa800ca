#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs produced by tools/profile_round.sh into a short text report
(kernel stats + per-kernel PMC sums/averages)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== bench line")
try:
    print(open(os.path.join(out, "bench.json")).read().strip())
except Exception as e:
    print("missing", e)

print("\n== rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
for f in find("stats/**/*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

pmc = defaultdict(lambda: defaultdict(list))
for f in find("pmc_*/**/*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "?")
            short = name.split("(")[0][-60:]
            pmc[short][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "0")))
print("\n== PMC per kernel (mean per dispatch; n dispatches)")
summary = {}
for k in pmc:
    if not any(t in k for t in ("count_kernel", "locate_kernel", "count_direct", "locate_walk", "count_tail", "plan_")):
        continue
    print(k)
    for c, v in sorted(pmc[k].items()):
        mean = sum(v) / len(v)
        print(f"   {c:24s} mean {mean:.6g}  n={len(v)}")
        summary.setdefault(k, {})[c] = mean
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
