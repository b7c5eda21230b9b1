#!/usr/bin/env python3
"""Phases of the detection launch's workgroups from UVO_HESS_STAMPS (100 MHz wall clock): python tools/probe/hess_stamps.py <csv>"""
import sys
import numpy as np
d = np.genfromtxt(sys.argv[1], delimiter=",", names=True)
t0 = d["t_start"].min()
span = (d["t_end"].max() - t0) / 100.0
print(f"launch span {span:.1f} us, {len(d)} workgroups")
for kind, name in ((0, "octave 0"), (1, "octave 1"), (2, "octave 2"), (3, "octave 3")):
    m = d["kind"] == kind
    if not m.any():
        continue
    life = (d["t_end"][m] - d["t_start"][m]) / 100.0
    fill = np.where(d["t_filled"][m] > 0, (d["t_filled"][m] - d["t_start"][m]) / 100.0, 0.0)
    det = (d["t_det"][m] - np.where(d["t_filled"][m] > 0, d["t_filled"][m], d["t_start"][m])) / 100.0
    nms = (d["t_end"][m] - d["t_det"][m]) / 100.0
    if "t_scan" in d.dtype.names:
        sc = (d["t_scan"][m] - d["t_det"][m]) / 100.0; sb = (d["t_scanbar"][m] - d["t_scan"][m]) / 100.0
        has = d["t_atomic"][m] > 0
        at = np.where(has, (d["t_atomic"][m] - d["t_scanbar"][m]) / 100.0, 0.0); wr = (d["t_end"][m] - np.where(has, d["t_atomic"][m], d["t_scanbar"][m])) / 100.0
        print(f"    nms split: scan {sc.mean():.2f}  barrier {sb.mean():.2f}  atomic+barrier {at[has].mean() if has.any() else 0:.2f} ({has.mean() * 100:.0f} % of tiles)  records+exit {wr.mean():.2f}")
    print(f"{name}: {m.sum():5d} tiles  life {life.mean():6.2f} us (p50 {np.median(life):.2f}, max {life.max():.2f})  fill {fill.mean():5.2f}  box sums {det.mean():5.2f}  nms {nms.mean():5.2f}   slot-time {life.sum() / 768:6.1f} us of the launch")
# how many workgroups are alive over time
ev = np.concatenate([np.c_[d["t_start"], np.ones(len(d))], np.c_[d["t_end"], -np.ones(len(d))]])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1]); t = (ev[:, 0] - t0) / 100.0
for q in (0.05, 0.25, 0.5, 0.75, 0.9, 0.97):
    i = np.searchsorted(t, q * span)
    print(f"  at {q * span:6.1f} us: {int(alive[min(i, len(alive) - 1)])} workgroups resident")
