"""Timeline of the C++ tracking loop (examples/icp_slam) from a rocprofv3
--kernel-trace CSV: where a frame's wall time goes, launch family by launch
family, with the idle time BEFORE each launch (end of the previous launch on
any stream -> start of this one) attributed to the family that follows it.

    python tools/slam_timeline.py <kernel_trace.csv> [frames]
"""
import csv
import sys
from collections import defaultdict

FAMILIES = [
    ("search", "SearchAccumulateKernel"), ("final_sum", "FinalSumKernel"),
    ("vds", "Vds"), ("vds", "SortHist"), ("vds", "SortScatter"),
    ("post_counts", "PostCounts"), ("index", "CountKernel"),
    ("index", "AssignRanges"), ("index", "::ScatterKernel"),
    ("ray_cast", "RayCastKernel"), ("range", "EstimateRange"),
    ("range", "RangeFill"), ("unproject", "UnprojectKernel"),
    ("integrate", "FrameStep"), ("integrate", "Integrate"),
    ("integrate", "Touch"), ("integrate", "Prepare"),
    ("transform", "Transform"), ("fill", "fill"), ("fill", "Fill"),
    ("copy", "copy"), ("copy", "Copy"),
]


def family(name):
    for f, pat in FAMILIES:
        if pat in name:
            return f
    return "other"


def main():
    path = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 59
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         r["Kernel_Name"]))
    rows.sort()
    # the loop: from the first search launch to the last kernel
    first = next(i for i, r in enumerate(rows) if "SearchAccumulate" in r[2])
    rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy = defaultdict(float)
    idle = defaultdict(float)
    count = defaultdict(int)
    gaps = defaultdict(list)
    prev_family = None
    frontier = rows[0][0]
    for st, en, name in rows:
        f = family(name)
        count[f] += 1
        if st > frontier:
            idle[f] += (st - frontier) / 1e3
            gaps[(prev_family, f)].append((st - frontier) / 1e3)
            frontier = st
        prev_family = f
        if en > frontier:
            busy[f] += (en - frontier) / 1e3
            frontier = en
    wall = (t1 - t0) / 1e3
    print("wall %.1f us per frame over %d frames" % (wall / frames, frames))
    print("%-12s %9s %14s %14s" % ("family", "launches", "busy us/frame",
                                   "idle-before us/frame"))
    for f in sorted(busy, key=lambda k: -(busy[k] + idle[k])):
        print("%-12s %9.1f %14.1f %14.1f" % (f, count[f] / frames,
                                             busy[f] / frames,
                                             idle[f] / frames))
    print("%-12s %9.1f %14.1f %14.1f" % (
        "total", sum(count.values()) / frames, sum(busy.values()) / frames,
        sum(idle.values()) / frames))
    print("idle gaps by (launch before -> launch after), us per frame; "
          "number per frame; median")
    for k in sorted(gaps, key=lambda k: -sum(gaps[k]))[:14]:
        g = sorted(gaps[k])
        print("  %-12s -> %-12s %7.1f %6.1f %7.1f" % (
            k[0], k[1], sum(g) / frames, len(g) / frames, g[len(g) // 2]))


if __name__ == "__main__":
    main()
