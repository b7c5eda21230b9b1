#!/usr/bin/env python3
"""Pipelined C3 rate of the 1st .. 5th context of ONE process, each destroyed before the next is created
("a context created after another one was destroyed runs ~10 % below its rate").
  UVO_STREAM_POOL=0 python tools/probe/ctx_reuse.py      # the runtime's own create / destroy
  python tools/probe/ctx_reuse.py                         # streams parked in the library's pool"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
import bench

W, H = bench.WIDTH, bench.HEIGHT
scene = synth.Scene(synth.SEEDS["C3"], W)
frames = [tuple(torch.from_numpy(a).cuda() for a in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rates = []
for it in range(5):
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=bench.MIN_HESSIAN_C3), 0, W, H, 8192)
    ctx.stereo_set_depth(6)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    order = bench.ping_pong(4)
    for _ in range(2):
        ctx.stereo_step(*frames[next(order)], 0.05)
    best = 0.0
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); sub = 0
        for i in range(n):
            while sub < n and sub - i < 6:
                ctx.stereo_submit(*frames[next(order)]); sub += 1
            ctx.stereo_collect(0.05)
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0))
    rates.append(best)
    ctx.close()
print("pool", os.environ.get("UVO_STREAM_POOL", "1"), "pairs/s of contexts 1..5:", " ".join(f"{r:.0f}" for r in rates), f"  5th/1st = {rates[4] / rates[0]:.3f}")
