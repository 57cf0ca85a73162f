#!/usr/bin/env python
"""VALU issue-rate calibration (tools/calib_valu.hip): per instruction stream
and occupancy, the SIMD cycles one wave64 instruction occupies (from the waves'
own clocks) and what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU read per issued
wave-instruction (from rocprofv3's counter pass of the same binary).

usage: summarize_valu.py <out dir of profile_valu.sh> <json out>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, dst = sys.argv[1], sys.argv[2]
alone = json.load(open(os.path.join(out, "alone.json")))
n_cu = alone["compute_units"]
n_simd = 4 * n_cu
N_XCD = 8

# counter pass: keyed by (template index, grid threads); the SECOND launch of
# each (the timed one) -- both run the same stream, so the mean is taken
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"),
                   recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"Stream<(\d+)>", r["Kernel_Name"])
        if not m:
            continue
        acc[(int(m.group(1)), int(r["Grid_Size"]))][r["Counter_Name"]].append(
            float(r["Counter_Value"]))
acc2 = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "sq2", "**", "*counter_collection.csv"),
                   recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"Stream<(\d+)>", r["Kernel_Name"])
        if not m:
            continue
        acc2[(int(m.group(1)), int(r["Grid_Size"]))][r["Counter_Name"]].append(
            float(r["Counter_Value"]))
names = []
for r in alone["runs"]:
    if r["stream"] not in names:
        names.append(r["stream"])

rows = []
for r in alone["runs"]:
    w = r["waves_per_simd"]
    # shader clock: s_memtime ticks per 100 MHz tick of s_memrealtime
    ticks = r["memtime_ticks_median"] / max(1.0, r["realtime_ticks_median"])
    row = {"stream": r["stream"], "waves_per_simd": w,
           "ns_per_wave_inst_per_simd": r["ns_per_wave_inst_per_simd"],
           "memtime_ticks_per_100mhz_tick": round(ticks, 3)}
    c = acc.get((names.index(r["stream"]), n_cu * w * 256))
    if c:
        m = {k: sum(v) / len(v) for k, v in c.items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        insts = m.get("SQ_INSTS_VALU", 0.0)
        row["counter_cycles"] = cyc
        row["SQ_INSTS_VALU"] = insts
        row["SQ_ACTIVE_INST_VALU"] = m.get("SQ_ACTIVE_INST_VALU")
        row["SQ_BUSY_CYCLES"] = m.get("SQ_BUSY_CYCLES")
        row["expected_wave_insts"] = alone["insts_per_wave"] * w * n_simd
        if insts and cyc:
            # issue cycles a wave-instruction occupies on its SIMD, taking the
            # whole launch as the denominator (includes launch ramp)
            row["simd_cycles_per_wave_inst"] = cyc * n_simd / insts
            row["ACTIVE_INST_VALU_per_inst"] = \
                m.get("SQ_ACTIVE_INST_VALU", 0.0) / insts
            # bench.py's frac_valu forms on a stream that IS the roof
            row["frac_valu_4cycle_form"] = \
                m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (n_simd * cyc)
            row["insts_per_simd_cycle"] = insts / (n_simd * cyc)
    c2 = acc2.get((names.index(r["stream"]), n_cu * w * 256))
    if c2:
        m2 = {k: sum(v) / len(v) for k, v in c2.items()}
        i2 = m2.get("SQ_INSTS_VALU", 0.0)
        for k in ("SQ_INST_CYCLES_VALU", "SQ_THREAD_CYCLES_VALU"):
            if k in m2 and i2:
                row[k + "_per_inst"] = m2[k] / i2
    rows.append(row)

# the roof bench.py uses: wave-instructions per SIMD per cycle of the plain
# float32 streams at the occupancy where they saturate
best = defaultdict(float)
for row in rows:
    if "insts_per_simd_cycle" in row:
        best[row["stream"]] = max(best[row["stream"]],
                                  row["insts_per_simd_cycle"])
res = {"device": alone["device"], "compute_units": n_cu,
       "peak_insts_per_simd_cycle": dict(best), "rows": rows}
json.dump(res, open(dst, "w"), indent=1)
for row in rows:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v)
                      for k, v in row.items()}))
print(json.dumps(res["peak_insts_per_simd_cycle"], indent=1))
