"""Print the newest rocprofv3 kernel_stats csv under gpurun_out/<dir>: python tools/probe/kstats.py <dir> [n]"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print("%-84s %5s %9.2f" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
