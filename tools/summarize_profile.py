#!/usr/bin/env python
"""Summarises rocprofv3 output of tools/profile_gpu.sh into small text/JSON
files that are committed under profiles/.

  <out>/trace/**/trace_kernel_stats.csv      -> <sum>/<tag>_kernel_stats.csv (verbatim, it is small)
  <out>/pmc_FETCH_SIZE/**/pmc_counter_collection.csv
  <out>/pmc_WRITE_SIZE/**/pmc_counter_collection.csv
                                             -> <sum>/<tag>_hbm_traffic.json (per-kernel mean per launch)

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
stream, so the read side is reported both raw and doubled ("corrected").
"""
import csv
import re
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def find(root, pattern):
    r = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return r[0] if r else None


def short(name):
    """Kernel base name: strips return type, namespaces, template and call
    arguments ("void o3dmi::(anonymous namespace)::Foo<int>(args)" -> "Foo")."""
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name.strip())
    name = re.split(r"[<(]", name, maxsplit=1)[0]
    return name.split("::")[-1].strip()


def main():
    out, summ, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(summ, exist_ok=True)
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(summ, "%s_kernel_stats.csv" % tag))
        with open(stats) as f:
            rows = list(csv.DictReader(f))
        print("kernel stats (top 12):")
        for r in rows[:12]:
            print("  %-60s calls %6s avg_ns %10s pct %s" % (
                short(r.get("Name", ""))[:60], r.get("Calls"),
                r.get("AverageNs"), r.get("Percentage")))
    traffic = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        p = find(os.path.join(out, "pmc_" + cname), "*counter_collection.csv")
        if not p:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        with open(p) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != cname:
                    continue
                k = short(r.get("Kernel_Name", ""))
                a = acc[k]
                a[0] += float(r.get("Counter_Value", 0))
                a[1] += 1
        for k, (s, n) in acc.items():
            traffic.setdefault(k, {})[cname + "_KiB_per_launch"] = s / max(n, 1)
            traffic[k]["launches_" + cname] = n
    for k, d in traffic.items():
        f = d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
        w = d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
        d["read_bytes_raw"] = f
        d["read_bytes_corrected_x2"] = 2 * f
        d["write_bytes"] = w
        d["hbm_bytes_per_launch_corrected"] = 2 * f + w
    with open(os.path.join(summ, "%s_hbm_traffic.json" % tag), "w") as fo:
        json.dump(traffic, fo, indent=1, sort_keys=True)
    for k, d in sorted(traffic.items()):
        print("traffic %-40s %s" % (k[:40], {a: round(b) for a, b in d.items()
                                              if "bytes" in a}))


if __name__ == "__main__":
    main()
