#!/usr/bin/env python3
"""Phase Gantt from a UVO_TRACE CSV (device timestamps, no profiler): one row per lane and stage.
  UVO_TRACE=gpurun_out/trace.csv python tools/prof_pipeline.py 300 ; python tools/probe/trace_view.py gpurun_out/trace.csv [first_pair] [n_pairs]"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 14
R = [(int(r["pair"]), int(r["lane"]), [float(r[k]) for k in ("a_begin_ms", "detect_end_ms", "a_end_ms", "b_begin_ms", "b_scored_ms", "b_end_ms")]) for r in rows]
R.sort()
allp = [r for r in R if r[2][5] > 0]
if len(allp) > 20:
    p = allp[10:-5]
    ends = np.array([r[2][5] for r in p]); begins = np.array([r[2][0] for r in p])
    print(f"pairs {p[0][0]}..{p[-1][0]}: cadence {np.median(np.diff(sorted(ends)))*1e3:.0f} us/pair (median), latency A begin -> B end {np.median(ends - begins)*1e3:.0f} us")
    d = np.array([[r[2][1] - r[2][0], r[2][2] - r[2][1], r[2][3] - r[2][2], r[2][4] - r[2][3], r[2][5] - r[2][4]] for r in p]) * 1e3
    print("median us: detection %.0f | tail (match..extract3d) %.0f | wait for worker/slot %.0f | hypotheses+score %.0f | host scan + mask + refit %.0f" % tuple(np.median(d, 0)))
sel = [r for r in R if first <= r[0] < first + npairs]
t0 = min(r[2][0] for r in sel); t1 = max(max(r[2]) for r in sel)
W = 180
print(f"pairs {first}..{first+npairs-1}: {t1 - t0:.3f} ms, {(t1 - t0) * 1e3 / W:.1f} us per column; D detection, t tail, . waiting, H hypotheses, R scan+refit")
lanes = sorted(set(r[1] for r in sel))
for ln in lanes:
    for stage in ("A", "B"):
        line = [" "] * (W + 1)
        for pr, l, t in sel:
            if l != ln: continue
            segs = [(t[0], t[1], "D"), (t[1], t[2], "t")] if stage == "A" else [(t[2], t[3], "."), (t[3], t[4], "H"), (t[4], t[5], "R")]
            for a, b, ch in segs:
                if b < 0 or a < 0: continue
                for x in range(int((a - t0) / (t1 - t0) * W), int((b - t0) / (t1 - t0) * W) + 1): line[min(x, W)] = ch
        print(f"lane {ln} {stage} |" + "".join(line) + "|")
