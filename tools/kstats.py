"""Print the top rows of a rocprofv3 kernel_stats.csv (name shortened)."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (total / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = r["Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    name = re.sub(r"^(o3dmi::|at::native::)", "", name)[:56]
    print("%-56s calls %5s avg %9.1f us  tot %8.2f ms" % (
        name, r["Calls"], float(r["AverageNs"]) / 1e3,
        float(r["TotalDurationNs"]) / 1e6))
