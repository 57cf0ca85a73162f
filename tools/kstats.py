"""Print the top rows of a rocprofv3 kernel_stats.csv (name shortened)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = r["Name"].split("(")[0].split("::")[-1][:48]
    print("%-48s calls %5s avg %10.1f us  %6s %%" % (
        name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
