#!/usr/bin/env python
"""Summaries of a rocprofv3 output directory for the FrameStepKernel launches
of one run (tools/c4_profile.sh):

    c4_summarize.py trace <dir>   per-launch durations / gaps from the kernel trace
    c4_summarize.py pmc <dir>     counters per launch + the kernel time of that pass
"""
import csv
import glob
import json
import os
import sys

KERNEL = os.environ.get("O3DMI_SUMMARIZE_KERNEL", "FrameStepKernel")


def _rows(d, pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            out.extend(csv.DictReader(fh))
    return out


def _grid(row):
    for k in ("Grid_Size", "Grid_Size_X", "Workgroup_Count"):
        if k in row and row[k]:
            try:
                return int(row[k])
            except ValueError:
                pass
    return 0


def trace(d):
    rows = [r for r in _rows(d, "*kernel_trace.csv") if KERNEL in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if not rows:
        return {"error": "no %s launches in %s" % (KERNEL, d)}
    t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), _grid(r))
         for r in rows]
    dur = [(b - a) / 1e3 for a, b, _ in t]
    gaps = [(t[i + 1][0] - t[i][1]) / 1e3 for i in range(len(t) - 1)]
    gs = sorted(gaps) if gaps else [0.0]
    # the un-fused run alternates front-only / integrate-only launches: the
    # integrate launches are the ones with the larger grid
    grids = [g for _, _, g in t]
    return {"launches": len(t), "span_ms": (t[-1][1] - t[0][0]) / 1e6,
            "kernel_ms_sum": sum(dur) / 1e3,
            "avg_us": sum(dur) / len(dur),
            "gap_us_p50": gs[len(gs) // 2], "gap_us_p90": gs[int(len(gs) * 0.9)],
            "gap_us_sum": sum(gaps),
            "per_launch_us": [round(x, 1) for x in dur],
            "per_launch_grid": grids,
            "per_launch_gap_us": [round(x, 1) for x in gaps]}


def pmc(d):
    acc, disp = {}, {}
    for r in _rows(d, "*counter_collection.csv"):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + \
            float(r["Counter_Value"])
        disp.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
    n = max([len(v) for v in disp.values()] or [1])
    tr = trace(d)
    out = {"launches": n, "per_launch": {k: v / n for k, v in acc.items()},
           "total": acc}
    if "error" not in tr:
        out["kernel_us_avg_this_pass"] = tr["avg_us"]
        out["kernel_ms_sum_this_pass"] = tr["kernel_ms_sum"]
    return out


if __name__ == "__main__":
    print(json.dumps({"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2])))
