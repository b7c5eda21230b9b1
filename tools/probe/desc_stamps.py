#!/usr/bin/env python3
"""Phases of the descriptor launch's small-window workgroups from UVO_DESC_STAMPS (100 MHz wall clock): python tools/probe/desc_stamps.py <csv>"""
import sys
import numpy as np
d = np.genfromtxt(sys.argv[1], delimiter=",", names=True)
d = d[d["t_end"] > 0]
if len(d) == 0:
    d = np.zeros(1, d.dtype)
t0 = d["t_start"].min()
span = (d["t_end"].max() - t0) / 100.0
print(f"span of the small-window workgroups {span:.1f} us, {len(d)} keypoints")
us = lambda a, b: (d[a] - d[b]) / 100.0
for kind, name in ((0, "general scale (resizeArea_)"), (2, "integer scale (resizeAreaFast_)")):
    m = d["kind"] == kind
    if not m.any():
        continue
    life = us("t_end", "t_start")[m]
    if kind == 2:
        print(f"{name}: {m.sum()} keypoints  life {life.mean():.2f} us  kp load {us('t_kp', 't_start')[m].mean():.2f}  resize {us('t_vert', 't_kp')[m].mean():.2f}  tail {us('t_end', 't_vert')[m].mean():.2f}")
        continue
    print(f"{name}: {m.sum()} keypoints  life {life.mean():.2f} us (p50 {np.median(life):.2f} max {life.max():.2f})  kp load {us('t_kp', 't_start')[m].mean():.2f}  "
          f"table/stage+barrier {us('t_ready', 't_kp')[m].mean():.2f}  horizontal {us('t_horiz', 't_ready')[m].mean():.2f}  vertical {us('t_vert', 't_horiz')[m].mean():.2f}  tail {us('t_end', 't_vert')[m].mean():.2f}   window {d['win'][m].mean():.0f} px")
ev = np.concatenate([np.c_[d["t_start"], np.ones(len(d))], np.c_[d["t_end"], -np.ones(len(d))]])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1]); t = (ev[:, 0] - t0) / 100.0
for q in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = np.searchsorted(t, q * span)
    print(f"  at {q * span:6.1f} us: {int(alive[min(i, len(alive) - 1)])} keypoint workgroups resident")

import os
if os.path.exists(sys.argv[1] + ".big"):
    b = np.genfromtxt(sys.argv[1] + ".big", delimiter=",", names=True)
    tb0 = b["t_start"].min()
    print(f"large-window tasks: {len(b)}, span {(b['t_end'].max() - tb0) / 100.0:.1f} us")
    for kind, name in ((1, "one column (window > 246)"), (3, "three columns (129..246)"), (11, "one column, integer scale"), (13, "three columns, integer scale")):
        m = b["kind"] == kind
        if not m.any():
            continue
        life = (b["t_end"][m] - b["t_start"][m]) / 100.0
        hz = np.where(b["t_horiz"][m] > 0, (b["t_horiz"][m] - b["t_start"][m]) / 100.0, np.nan)
        print(f"  {name}: {m.sum()} tasks  life {life.mean():.2f} us (p50 {np.median(life):.2f} max {life.max():.2f})  horizontal {np.nanmean(hz):.2f}  vertical+store {life.mean() - np.nanmean(hz):.2f}  window {b['win'][m].mean():.0f} px")
    ev = np.concatenate([np.c_[b["t_start"], np.ones(len(b))], np.c_[b["t_end"], -np.ones(len(b))]])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1]); t = (ev[:, 0] - tb0) / 100.0
    sp = (b["t_end"].max() - tb0) / 100.0
    for q in (0.1, 0.3, 0.5, 0.7, 0.9):
        i = np.searchsorted(t, q * sp)
        print(f"  at {q * sp:6.1f} us: {int(alive[min(i, len(alive) - 1)])} tasks (waves) running")
