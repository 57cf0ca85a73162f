"""Sum rocprofv3 --pmc counter_collection.csv per kernel (name shortened)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name).split("(")[0]
        name = re.sub(r"^(o3dmi::|at::native::)", "", name)
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
out = {}
for k, v in acc.items():
    if not any(x in k for x in sys.argv[3:] or [""]):
        continue
    n = max(1, len(calls[k]))
    out[k] = {"dispatches": n, **{c: val / n for c, val in sorted(v.items())}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
