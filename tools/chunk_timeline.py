#!/usr/bin/env python
"""Per-work-item timeline of ONE ChunkIntegrateKernel launch (sliced path),
recorded by the library when O3DMI_CHUNK_TIMELINE=<file> is set
(vbg_stream.hip, LaunchChunkIntegrate): what a rank's chunk launch lasts as
long as -- the slowest compute unit's sum of work, or the longest item's chain.

    chunk_timeline.py <file> [--json]

Record per (entry, part): start / end on the 100 MHz clock, the entry's frame
count, HW_ID (wave, SIMD, CU, SE) and the XCD.
"""
import json
import sys

import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint64)
    cap, parts, n_frames, grid = [int(x) for x in raw[:4].astype(np.int64)]
    rec = raw[4:].reshape(-1, 4)
    rec = rec[rec[:, 1] != 0]
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    n_set = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    hw = (rec[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcd = (rec[:, 3] >> np.uint64(32)).astype(np.int64)
    wg = (rec[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    cu_key = ((xcd * 8 + se) * 2 + sh) * 16 + cu
    base = t0.min()
    return {"cap": cap, "parts": parts, "n_frames": n_frames, "grid": grid,
            "t0": (t0 - base) / 100.0, "t1": (t1 - base) / 100.0,  # us
            "n_set": n_set, "cu": cu_key, "xcd": xcd, "wg": wg}


def analyse(d):
    t0, t1, n = d["t0"], d["t1"], d["n_set"]
    dur = t1 - t0
    span = float(t1.max())
    out = {"items": int(len(t0)), "n_frames": d["n_frames"], "grid": d["grid"],
           "span_us": span,
           "frames_per_item_mean": float(n.mean()),
           "frames_per_item_max": int(n.max()),
           "item_us_mean": float(dur.mean()), "item_us_max": float(dur.max()),
           "start_us_p50": float(np.median(t0)), "start_us_max": float(t0.max())}
    # items resident over time (the machine holds 256 CUs x k workgroups)
    edges = np.linspace(0, span, 21)
    resident = [int(((t0 <= e) & (t1 > e)).sum()) for e in edges[:-1]]
    out["resident_workgroups_at_5pct_steps"] = resident
    # per frame-count class: time per frame
    cls = []
    for lo, hi in ((1, 8), (9, 32), (33, 96), (97, 160), (161, 256)):
        m = (n >= lo) & (n <= hi)
        if m.any():
            cls.append({"frames": "%d-%d" % (lo, hi), "items": int(m.sum()),
                        "item_us_mean": float(dur[m].mean()),
                        "us_per_frame": float((dur[m] / n[m]).mean()),
                        "start_us_mean": float(t0[m].mean()),
                        "end_us_mean": float(t1[m].mean()),
                        "end_us_max": float(t1[m].max())})
    out["by_frame_count"] = cls
    # per compute unit: when it went idle, how much work (frames) it got
    cus = np.unique(d["cu"])
    end = np.array([t1[d["cu"] == c].max() for c in cus])
    work = np.array([n[d["cu"] == c].sum() for c in cus])
    items = np.array([(d["cu"] == c).sum() for c in cus])
    out["compute_units_seen"] = int(len(cus))
    out["cu_end_us"] = {"min": float(end.min()), "p50": float(np.median(end)),
                        "p90": float(np.percentile(end, 90)),
                        "max": float(end.max())}
    out["cu_work_frames"] = {"min": int(work.min()), "mean": float(work.mean()),
                             "max": int(work.max())}
    out["cu_items"] = {"min": int(items.min()), "mean": float(items.mean()),
                       "max": int(items.max())}
    out["corr_cu_end_vs_work"] = float(np.corrcoef(end, work)[0, 1])
    # the last items to finish: who were they
    last = np.argsort(t1)[-8:]
    out["last_items"] = [{"frames": int(n[i]), "start_us": float(t0[i]),
                          "end_us": float(t1[i]), "wg": int(d["wg"][i])}
                         for i in last]
    return out


if __name__ == "__main__":
    res = analyse(load(sys.argv[1]))
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        print(json.dumps(res, indent=1))
