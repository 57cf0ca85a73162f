#!/usr/bin/env python
"""o3dmi kernels of a rocprofv3 kernel trace: per-kernel statistics and a window
from the last third of the run (start us, duration us, grid, kernel): what
overlaps what.   trace_window.py <kernel_trace.csv> [rows]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "o3dmi" not in n and "rocclr" not in n and "nccl" not in n.lower():
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     r.get("Grid_Size", r.get("Grid_Size_X", "")),
                     n.split("(")[0].split("::")[-1][:70]))
rows.sort()
stat = {}
for a, b, g, n in rows:
    s = stat.setdefault(n, [0, 0.0, 1e18, 0.0])
    d = (b - a) / 1e3
    s[0] += 1
    s[1] += d
    s[2] = min(s[2], d)
    s[3] = max(s[3], d)
for n, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print("%-72s n %6d avg %8.1f min %8.1f max %8.1f us" % (n, s[0], s[1] / s[0], s[2], s[3]))
print()
n = len(rows)
lo = max(0, n * 2 // 3)
t0 = rows[lo][0]
m = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for a, b, g, name in rows[lo:lo + m]:
    print("%10.1f %8.1f %9s %s" % ((a - t0) / 1e3, (b - a) / 1e3, g, name))
