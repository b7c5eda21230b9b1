#!/usr/bin/env python3
"""HIP-event time of the detection launch (stage hessian_nms_o0) and of the whole synchronous step at C3, for A/B of launch shapes:
   UVO_HESS_P=384,160,224 python tools/probe/hess_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
for i in range(6):
    ctx.stereo_step(*dev[order[i % 6]], 0.05)
lat = []
for i in range(60):
    a = time.perf_counter(); r = ctx.stereo_step(*dev[order[i % 6]], 0.05); lat.append((time.perf_counter() - a) * 1e3)
ctx.timing_enable(True); ctx.timing_reset()
for i in range(12):
    r = ctx.stereo_step(*dev[order[i % 6]], 0.05)
tm = ctx.timing()
ms, n = tm["hessian_nms_o0"]
lat.sort()
print(f"UVO_HESS_P={os.environ.get('UVO_HESS_P', 'default'):>14}  hessian launch {ms / n * 1e3:7.1f} us   step median {lat[len(lat) // 2]:.4f} ms   kpts {r.n_left} inliers {r.n_inliers}")
ctx.close()
