"""Per-kernel averages of the counters in gpurun_out/<dir>/**/*_counter_collection.csv: python tools/probe/pmc_mix.py <dir>"""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("%-60s" % "kernel", " ".join("%16s" % n[-16:] for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
    print("%-60s" % k, " ".join("%16.0f" % (sum(v[n]) / len(v[n]) if v.get(n) else -1) for n in names))
