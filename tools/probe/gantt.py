#!/usr/bin/env python3
"""Text Gantt chart of a window of a rocprofv3 kernel trace, one row per HIP stream/queue.  python tools/probe/gantt.py DIR [start_ms] [len_ms]"""
import csv, glob, sys
d = sys.argv[1]; t_start = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0; t_len = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[0]
rows = [(r["Kernel_Name"].replace("uvo::", "").replace("void ", "").split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in csv.DictReader(open(f))]
hs = sorted(r[1] for r in rows if r[0].startswith("k_hessian_nms_c<0"))
# origin: first o0 launch of the densest stretch
import numpy as np
g = np.diff(hs); i0 = int(np.argmax(np.convolve((g < 600_000).astype(int), np.ones(50), "valid")))
t0 = hs[i0] + int(t_start * 1e6) * 0 + int(t_start * 1e6) - int(30e6) + 0 if False else hs[i0] + int(t_start * 1e6)
t1 = t0 + int(t_len * 1e6)
code = {"k_integral": "i", "k_hessian_nms_c<0": "0", "k_hessian_nms_c<1": "1", "k_hessian_nms_p<2": "2", "k_hessian_nms_p<3": "3", "k_hessian_finish": "f", "k_rank": "r",
        "k_big_sort": "s", "k_descriptor64_big_tabs": "t", "k_descriptor64_big_finish": "e", "k_descriptor64(": "D", "k_match_mfma": "M",
        "k_match_resolve": "m", "k_match_compact": "c", "k_gather": "g", "k_triangulate": "T", "k_extract3d": "x", "k_pnp_hyp": "H", "k_pnp_score": "S", "k_pnp_mask": "k",
        "k_pnp_refit": "R", "__amd": "u"}
W = 200
qs = sorted(set(r[3] for r in rows if t0 <= r[1] < t1), key=int)
print(f"window {t_len} ms, {t_len * 1e3 / W:.1f} us per column; legend: " + " ".join(f"{v}={k}" for k, v in code.items()))
for q in qs:
    line = [" "] * W
    for n, s, e, qq in rows:
        if qq != q or e < t0 or s >= t1: continue
        ch = next((v for k, v in code.items() if n.startswith(k)), "?")
        a = max(0, int((s - t0) / (t1 - t0) * W)); b = min(W - 1, int((e - t0) / (t1 - t0) * W))
        for x in range(a, b + 1): line[x] = ch
    print(f"q{q:>3s} |" + "".join(line) + "|")
