#!/usr/bin/env python3
"""tools/trace_gaps.py <dir with rocprofv3 --kernel-trace csv>: the last steps of a bench run as a timeline --
every kernel's duration and the idle gap before it (same queue), to see what a step spends outside its kernels."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = [i for i, r in enumerate(rows) if "femto_amd::count_direct_kernel" in r["Kernel_Name"] or "femto_amd::count_kernel" in r["Kernel_Name"]]
lo = steps[-n] if len(steps) >= n else 0
tail = rows[max(0, lo - 2):steps[-1] + 5] if steps else rows[-24:]     # the last n timed steps
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("gap %8.1f us  dur %9.1f us  %s" % (gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
