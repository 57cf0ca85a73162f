#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE calibration (tools/calib_hbm.hip): expected bytes of
each calibration kernel / counter bytes (KiB x 1024) -> correction factors.

usage: summarize_calib.py <out dir of profile_step_pmc.sh> <json out>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, dst = sys.argv[1], sys.argv[2]
expected = None
for log in ("calib_FETCH_SIZE.log", "calib_WRITE_SIZE.log"):
    p = os.path.join(out, log)
    if os.path.exists(p):
        txt = open(p).read()
        i = txt.find('{"ReadKernel<1>"')
        if i >= 0:
            j = txt.find("}}", i)
            expected = json.loads(txt[i:j + 2])
            break
if expected is None:
    sys.exit("no calibration log")


def short(name):
    name = re.sub(r"^void\s+", "", name.strip())
    return name.split("(")[0]


res = {k: dict(v) for k, v in expected.items()}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, "calib_" + cname, "**",
                                    "*counter_collection.csv"),
                       recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k, (s, n) in acc.items():
        if k in res and n:
            res[k][cname + "_bytes_per_launch"] = s / n * 1024.0
# kernel durations
for f in glob.glob(os.path.join(out, "calib_trace", "**", "*kernel_stats.csv"),
                   recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k in res:
            res[k]["avg_ns"] = float(r["AverageNs"])
for k, v in res.items():
    f, w = v.get("FETCH_SIZE_bytes_per_launch"), v.get("WRITE_SIZE_bytes_per_launch")
    if f and v["read"]:
        v["read_factor"] = v["read"] / f      # multiply FETCH_SIZE by this
    if w and v["write"]:
        v["write_factor"] = v["write"] / w
    if v.get("avg_ns"):
        v["GBps"] = (v["read"] + v["write"]) / v["avg_ns"]
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
