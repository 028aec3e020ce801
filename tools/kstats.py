"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir-or-csv> [n]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for i, r in enumerate(csv.DictReader(open(p))):
    if i >= n:
        break
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-70s calls %5s  avg %9.1f us  %5s %%" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
