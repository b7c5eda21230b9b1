#!/usr/bin/env python3
"""(Round 1; the round-2 pass is tools/probe/final_profiles_r02.sh + tools/pmc_summary_r02.py.)  Summarise rocprofv3 output into profiles/: per-kernel time table (kernel_stats.csv) and the HBM
traffic of the dominant kernel from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes
(MI355X_MICROARCH.md 'HBM': both counters are in KiB; on gfx950 FETCH_SIZE reads exactly half the
bytes of a wide coalesced stream, other access widths are uncalibrated -> raw and x2 are both kept)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KERNEL = "k_hessian_nms_c<0"


def counter_avg(dirname, counter):
    f = sorted(glob.glob(os.path.join(OUT, dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return None, 0
    vals = []
    for r in csv.DictReader(open(f[0])):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals) if vals else None), len(vals)


def main(tag):
    fetch, nf = counter_avg("pmc_fetch", "FETCH_SIZE")
    write, nw = counter_avg("pmc_write", "WRITE_SIZE")
    res = {"kernel": "uvo::k_hessian_nms_c<0, 64, 24, 512> (octave 0, both images of a pair per launch)",
           "launches_sampled": [nf, nw], "FETCH_SIZE_KiB_avg": fetch, "WRITE_SIZE_KiB_avg": write}
    if fetch is not None and write is not None:
        res["hbm_bytes_per_launch_raw"] = int((fetch + write) * 1024)
        res["hbm_bytes_per_launch"] = int((2 * fetch + write) * 1024)       # gfx950 FETCH_SIZE x2 correction
        res["note"] = ("FETCH_SIZE doubled per the gfx950 correction for coalesced streams; this kernel reads 4 B/lane "
                       "tile rows, an uncalibrated width, so the true read traffic lies between raw and corrected")
    json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_hessian_o0.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
    ks = sorted(glob.glob(os.path.join(OUT, "prof_%s" % tag, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag), "w") as g:
            g.write(open(ks[0]).read())
        with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.md" % tag), "w") as g:
            g.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
            for r in rows:
                g.write("| %s | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                    float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        for r in rows[:14]:
            print("%-70s %6s %9.2f us %s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
