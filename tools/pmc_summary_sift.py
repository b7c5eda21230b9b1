#!/usr/bin/env python3
"""gpurun_out/pmc_sift_* (tools/probe/gpu.sh sift-pmc) -> profiles/r03_pmc_sift_kernels.json: per-launch counter means of the SIFT
kernels on the 1080p frames (for the tiled blur: the octave-0 launch of each radius), HBM bytes per MI355X_MICROARCH.md's recipe
(FETCH_SIZE / WRITE_SIZE in KiB, separate passes; FETCH doubled for coalesced streams on gfx950, raw kept beside it) and the ratios
DESIGN.md quotes."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KERNELS = ["k_sift_blur_tile<5", "k_sift_blur_tile<6", "k_sift_blur_tile<8", "k_sift_blur_tile<10", "k_sift_blur_tile<13", "k_sift_descriptor", "k_sift_orient",
           "k_sift_extrema_all", "k_sift_rank", "k_sift_tail", "k_sift_refine", "k_sift_resize2x"]
SIMDS, GHZ = 1024, 2.1


def one_pass(d):
    f = sorted(glob.glob(os.path.join(OUT, d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    disp = {}
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]) if "Grid_Size" in r else 0, "c": defaultdict(float),
                                "ns": float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for key in KERNELS:
        sel = [disp[i] for i in sorted(disp) if key in disp[i]["name"]]
        if not sel:
            continue
        gmax = max(e["grid"] for e in sel)
        if "blur_tile" in key:
            sel = [e for e in sel if e["grid"] == gmax][1:] or sel  # the octave-0 launches of the 1080p frames (largest grid), the first one (cold) dropped
        else:
            sel = sel[1:len(sel) // 2]                                # one launch per frame: prof_sift.py runs its 1080p frames first, then as many at 640 x 360
        acc = defaultdict(float)
        for e in sel:
            for c, v in e["c"].items():
                acc[c] += v / len(sel)
        out[key] = {"launches": len(sel), "grid_threads": gmax, "avg_us_in_pass": round(sum(e["ns"] for e in sel) / len(sel) / 1e3, 2), **{c: round(v, 1) for c, v in acc.items()}}
    return out


res = defaultdict(dict)
for d in ("pmc_sift_fetch", "pmc_sift_write", "pmc_sift_sq1", "pmc_sift_sq2"):
    for k, v in one_pass(d).items():
        for c, x in v.items():
            if c == "avg_us_in_pass":
                res[k].setdefault("avg_us_by_pass", {})[d] = x
            else:
                res[k][c] = x
for k, v in res.items():
    us = v["avg_us_by_pass"].get("pmc_sift_sq1", 0)
    der = {}
    if "FETCH_SIZE" in v:
        der["hbm_bytes_raw"] = int((v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0)) * 1024)
        der["hbm_bytes_fetch_x2"] = int((2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0)) * 1024)
    if us and "SQ_ACTIVE_INST_VALU" in v:
        der["simd_valu_busy_at_2.1GHz"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (us * 1e-6 * GHZ * 1e9), 4)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_cycles_per_active_cycle"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
    if v.get("SQ_WAVE_CYCLES"):
        der["valu_issue_share_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4)
        if "SQ_WAIT_ANY" in v:
            der["parked_share (s_waitcnt / barrier)"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
        if "SQ_WAIT_INST_ANY" in v:
            der["issue_stall_share"] = round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4)
    if v.get("SQ_INSTS_VALU"):
        der["salu_per_valu"] = round(v.get("SQ_INSTS_SALU", 0) / v["SQ_INSTS_VALU"], 3)
    v["derived"] = der
json.dump({"source": "tools/probe/gpu.sh sift-pmc: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/prof_sift.py 4 (one pass per group)",
           "units": "counter means per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB",
           "kernels": res}, open(os.path.join(ROOT, "profiles", "r03_pmc_sift_kernels.json"), "w"), indent=1)
for k, v in res.items():
    print(k, v.get("avg_us_by_pass"), v["derived"])
