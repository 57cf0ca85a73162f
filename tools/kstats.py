"""Per-kernel statistics of a rocprofv3 --kernel-trace run (its rocpd .db).

  python tools/kstats.py <results.db> [--frames N] [--csv out.csv]

rocprofv3 7.x writes a SQLite database instead of the *_kernel_stats.csv of
earlier releases; this prints (and optionally writes, in that CSV's columns)
calls, total / average / min / max duration per kernel name, and with
--frames the per-frame launch count and kernel time.
"""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
    n = name.replace("o3dmi::(anonymous namespace)::", "").replace(
        "(anonymous namespace)::", "").replace("o3dmi::", "")
    n = re.sub(r"^void ", "", n)
    depth = 0
    for i, ch in enumerate(n):          # cut the argument list, keep <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--frames", type=float, default=0)
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute(
        "select name, count(*), sum(end-start), avg(end-start), "
        "min(end-start), max(end-start) from kernels group by name "
        "order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    print("kernels: %d launches, %.1f us of kernels, %.1f us first start -> "
          "last end" % (sum(r[1] for r in rows), total / 1e3,
                        (span[1] - span[0]) / 1e3))
    if a.frames:
        print("per frame (%g frames): %.2f launches, %.1f us of kernels" %
              (a.frames, sum(r[1] for r in rows) / a.frames,
               total / 1e3 / a.frames))
    for r in rows[:a.top]:
        line = "%-64s n=%6d avg %8.2f us  total %10.1f us %5.1f%%" % (
            short(r[0])[:64], r[1], r[3] / 1e3, r[2] / 1e3,
            100.0 * r[2] / total)
        if a.frames:
            line += "  | per frame %5.2f x, %7.1f us" % (
                r[1] / a.frames, r[2] / 1e3 / a.frames)
        print(line)
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs",
                        "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], round(r[3], 3),
                            round(100.0 * r[2] / total, 4), r[4], r[5]])
    return 0


if __name__ == "__main__":
    sys.exit(main())
