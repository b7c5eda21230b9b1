#!/usr/bin/env python3
"""Pipeline timeline from a rocprofv3 --kernel-trace CSV: per kernel, duration while the pipelined region of bench.py runs vs
alone (the synchronous latency leg), wall-clock share, and how many detection kernels are in flight at once.
  python tools/probe/timeline.py gpurun_out/<dir>"""
import csv, glob, sys, collections
import numpy as np

def short(n):
    n = n.replace("uvo::", "").replace("void ", "")
    return n.split("(")[0][:44]

def main(d):
    f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[0]
    rows = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])) for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r[1])
    hs = [r for r in rows if r[0].startswith("k_hessian_nms_c<0")]
    st = np.array([r[1] for r in hs])
    gaps = np.diff(st)
    # pipelined region: consecutive o0 launches closer than 0.6 ms; the synchronous legs are ~0.9 ms apart
    dense = gaps < 600_000
    # longest dense run(s)
    runs, i = [], 0
    while i < len(dense):
        if dense[i]:
            j = i
            while j < len(dense) and dense[j]: j += 1
            runs.append((i, j)); i = j
        else: i += 1
    runs.sort(key=lambda r: r[0] - r[1])
    for name, (a, b) in zip(("pipelined region A", "pipelined region B (host frames)"), runs[:2]):
        t0, t1 = st[a], hs[b][2]
        sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
        wall = (t1 - t0) / 1e3
        npairs = b - a + 1
        print(f"== {name}: {npairs} pairs in {wall/1e3:.2f} ms -> {npairs / wall * 1e6:.0f} pairs/s")
        agg = collections.defaultdict(list)
        for r in sel: agg[r[0]].append((r[2] - r[1]) / 1e3)
        tot = sum(sum(v) for v in agg.values())
        print(f"   sum of kernel durations / wall = {tot / wall:.2f} (average kernels in flight)")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
            print(f"   {k:46s} n/pair {len(v)/npairs:5.2f}  avg {np.mean(v):8.1f} us  share-of-wall {sum(v)/wall*100:6.1f} %")
        # detection kernels in flight
        det = [r for r in sel if r[0].startswith(("k_hessian", "k_integral", "k_descriptor", "k_rank", "k_big", "k_desc"))]
        ev = sorted([(r[1], 1) for r in det] + [(r[2], -1) for r in det])
        cur, last, hist = 0, t0, collections.Counter()
        for t, dlt in ev:
            hist[cur] += t - last; last = t; cur += dlt
        tt = sum(hist.values())
        print("   detection kernels in flight (share of time): " + ", ".join(f"{k}: {v/tt*100:.0f}%" for k, v in sorted(hist.items())))
    # alone: synchronous leg = the sparse part
    sparse = [i for i in range(len(gaps)) if not dense[i]]
    if sparse:
        a, b = sparse[len(sparse)//4], sparse[-1]
        t0, t1 = st[a], st[b]
        sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
        agg = collections.defaultdict(list)
        for r in sel: agg[r[0]].append((r[2] - r[1]) / 1e3)
        print("== synchronous legs (one pair in flight): avg us per kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
            print(f"   {k:46s} avg {np.mean(v):8.1f} us")

if __name__ == "__main__":
    main(sys.argv[1])
