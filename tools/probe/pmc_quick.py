"""Per-launch means of the counters in a rocprofv3 --pmc output directory, for kernels whose name contains a pattern:
python tools/probe/pmc_quick.py <gpurun_out subdir> <pattern> [skip_first_n]"""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, pat = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True)[0]
per = defaultdict(lambda: defaultdict(float)); order = []
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:50], r["Dispatch_Id"])
    if k not in per:
        order.append(k)
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
tot = defaultdict(float); n = 0
for k in order[skip:]:
    n += 1
    for c, v in per[k].items():
        tot[c] += v
print(d, pat, "launches", n)
for c in sorted(tot):
    print("  %-28s %14.0f" % (c, tot[c] / max(n, 1)))
