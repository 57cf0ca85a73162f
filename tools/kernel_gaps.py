"""Inter-kernel gaps of one kernel family from a rocprofv3 --kernel-trace CSV:
how long the GPU sits between the end of one launch and the start of the next
(same stream, dependent launches).

    python tools/kernel_gaps.py <kernel_trace.csv> [name substring]
"""
import csv
import sys

import numpy as np


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "FrameStepKernel"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if pat in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                             r["Kernel_Name"]))
    rows.sort()
    st = np.array([r[0] for r in rows], dtype=np.int64)
    en = np.array([r[1] for r in rows], dtype=np.int64)
    dur = (en - st) / 1e3
    gap = (st[1:] - en[:-1]) / 1e3
    steady = gap[(gap < 200)]  # ignore batch boundaries / host waits
    print("launches %d  duration us: mean %.1f  p50 %.1f  p90 %.1f" % (
        len(rows), dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90)))
    print("gap to next launch us (n=%d of %d below 200): mean %.2f  p10 %.2f  "
          "p50 %.2f  p90 %.2f  max %.1f" % (
              steady.size, gap.size, steady.mean(), np.percentile(steady, 10),
              np.percentile(steady, 50), np.percentile(steady, 90),
              steady.max()))
    period = (st[1:] - st[:-1]) / 1e3
    period = period[period < 400]
    print("start-to-start us: mean %.1f p50 %.1f" % (period.mean(),
                                                     np.percentile(period, 50)))


if __name__ == "__main__":
    main()
