#!/usr/bin/env python3
"""python tools/pmc_summary.py [--round] rNN: gpurun_out/ of `TAG=rNN tools/probe/gpu.sh final rNN` -> profiles/rNN_*: kernel-time tables (pipelined bench run and synchronous run),
the roctx marker table, HBM bytes (FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md 'HBM': KiB units, FETCH_SIZE doubled for
coalesced streams on gfx950, other widths uncalibrated so raw and corrected are both kept) and the SQ counter passes of the
stage-A kernels with the ratios derived from them."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
T = ([a for a in sys.argv[1:] if a != "--round"] or ["r05"])[0]        # the round's tag (one script for every round; rounds 2-4 had a copy each)
KERNELS = {
    "hessian_all_octaves": "k_hessian_nms_all",
    "hessian_finish": "k_hessian_finish",
    "descriptor64": "k_descriptor64(",
    "integral_strip_final": "k_integral_strip_final",
    "rank_partial": "k_rank_partial",
    "match_mfma": "k_match_mfma",
    "match_resolve": "k_match_resolve",
    "pnp_hyp": "k_pnp_hyp",
    "pnp_refit_fast": "k_pnp_refit_fast",
}
SIMDS = 256 * 4


def newest(dirname, suffix):
    f = sorted(glob.glob(os.path.join(OUT, dirname, "**", "*" + suffix), recursive=True), key=os.path.getmtime, reverse=True)
    return f[0] if f else None


def counters(dirname):
    """{kernel key: {counter: mean per launch}}, {kernel key: mean duration ns} of one pass"""
    f = newest(dirname, "_counter_collection.csv")
    acc, dur = {}, {}
    if not f:
        return acc, dur
    seen = set()
    for r in csv.DictReader(open(f)):
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if (key, r["Dispatch_Id"]) not in seen:
                    seen.add((key, r["Dispatch_Id"]))
                    dur.setdefault(key, []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return ({k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()},
            {k: sum(v) / len(v) for k, v in dur.items()})


def stats_table(tag, dst):
    f = newest("prof_" + tag, "_kernel_stats.csv")
    if not f:
        return None
    rows = list(csv.DictReader(open(f)))
    shutil.copy(f, os.path.join(PROF, dst + ".csv"))
    with open(os.path.join(PROF, dst + ".md"), "w") as g:
        g.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows:
            g.write("| %s | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return {r["Name"]: float(r["AverageNs"]) for r in rows}


def main():
    os.makedirs(PROF, exist_ok=True)
    sync = stats_table(T + "_sync", T + "_final_sync_kernel_stats") or {}
    stats_table(T + "_pipe", T + "_final_kernel_stats")
    f = newest("prof_" + T + "_roctx", "_marker_api_stats.csv")
    if f:
        shutil.copy(f, os.path.join(PROF, T + "_roctx_marker_stats.csv"))
    for name in ("bench_%s_final.json" % T, "bench_%s_20steps.json" % T, "bench_%s_configs.json" % T, "bench_%s_1rank_rccl.json" % T):
        if os.path.exists(os.path.join(OUT, name)):
            shutil.copy(os.path.join(OUT, name), os.path.join(PROF, name.replace("bench_%s" % T, "%s_bench" % T)))
    fetch, _ = counters("pmc_%s_fetch" % T)
    write, _ = counters("pmc_%s_write" % T)
    sq1, d1 = counters("pmc_%s_sq1" % T)
    sq2, d2 = counters("pmc_%s_sq2" % T)
    mf, _ = counters("pmc_%s_mfma" % T)
    import hashlib
    sha = {f: hashlib.sha256(open(os.path.join(ROOT, "ergo_uvo_amd", "csrc", f), "rb").read()).hexdigest()[:16] for f in ("surf.hip", "match.hip", "pose.hip")}
    res = {"kernel_source_sha": sha, "source": "tools/probe/gpu.sh final (synchronous C3 run, tools/prof_stereo.py 6; one rocprofv3 --pmc pass per counter set); per-launch means",
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE / WRITE_SIZE in KiB",
           "kernels": {}}
    for key, pat in KERNELS.items():
        e = {}
        avg = [v for n, v in sync.items() if pat in n]
        if avg:
            e["avg_us_kernel_trace"] = round(avg[0] / 1e3, 2)
        if key in fetch and key in write:
            fk, wk = fetch[key].get("FETCH_SIZE"), write[key].get("WRITE_SIZE")
            e["FETCH_SIZE_KiB"] = fk; e["WRITE_SIZE_KiB"] = wk
            e["hbm_bytes_raw"] = int((fk + wk) * 1024); e["hbm_bytes_fetch_x2"] = int((2 * fk + wk) * 1024)
        c = dict(sq1.get(key, {})); c.update(sq2.get(key, {})); c.update(mf.get(key, {}))
        if c:
            e["counters"] = {k: round(v) for k, v in sorted(c.items())}
            wc = c.get("SQ_WAVE_CYCLES")
            if wc:
                der = {}
                for name, cn in (("valu_issue_share_of_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("lds_issue_share", "SQ_ACTIVE_INST_LDS"),
                                 ("any_issue_share", "SQ_ACTIVE_INST_ANY"), ("parked_share (s_waitcnt / barrier)", "SQ_WAIT_ANY"),
                                 ("issue_stall_share", "SQ_WAIT_INST_ANY")):
                    if cn in c:
                        der[name] = round(c[cn] / wc, 4)
                if "SQ_INSTS_SALU" in c and key in d2:
                    # one scalar unit per CU, one instruction per cycle: its share of the kernel's cycles on 256 CUs
                    der["scalar_unit_busy_at_2.1GHz"] = round(c["SQ_INSTS_SALU"] / 256 / (d2[key] * 2.1), 4)
                    if c.get("SQ_INSTS_VALU"):
                        der["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3)
                if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
                    der["lds_bank_conflict_cycles_per_active_cycle"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
                if "SQ_ACTIVE_INST_VALU" in c and key in d1:
                    # a wave64 VALU instruction holds its SIMD for one quad-cycle; 1024 SIMDs
                    der["simd_valu_busy_at_2.1GHz"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (d1[key] * 2.1), 4)
                    # the counter books every VALU instruction at one quad-cycle; conversions hold the SIMD for about two
                    # (tools/probe/issue_rate_probe.hip: v_cvt_f64_f32 / v_cvt_f32_f64 / v_cvt_f32_i32 at 1.7-1.8 x v_add_f32)
                    if "SQ_INSTS_VALU_CVT" in c:
                        der["simd_valu_busy_with_8_cycle_conversions"] = round((c["SQ_ACTIVE_INST_VALU"] + c["SQ_INSTS_VALU_CVT"]) * 4 / SIMDS / (d1[key] * 2.1), 4)
                    der["duration_us_in_the_counter_pass"] = round(d1[key] / 1e3, 2)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES"):
                    # SQ_VALU_MFMA_BUSY_CYCLES sums the four SIMDs' matrix pipes of a CU, SQ_BUSY_CU_CYCLES counts the CU once
                    der["mfma_busy_share_of_a_busy_cu_4_pipes"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 4)
                e["derived"] = der
        if e:
            res["kernels"][key] = e
    json.dump(res, open(os.path.join(PROF, T + "_pmc_stage_kernels.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    main()
