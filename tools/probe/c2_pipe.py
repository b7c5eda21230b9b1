"""Pipelined throughput at C2 (1280x720) or any size: python tools/probe/c2_pipe.py [width height min_hessian steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
mh = int(sys.argv[3]) if len(sys.argv) > 3 else 5685
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
scene = synth.Scene(synth.SEEDS["C2"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=mh), 0, W, H, 8192)
ctx.stereo_set_depth(6)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
for n in (60, steps):
    sub = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        while sub < n and sub - i < 6:
            ctx.stereo_submit(*dev[order[sub % 6]]); sub += 1
        r = ctx.stereo_collect(0.05)
    dt = time.perf_counter() - t0
print("%dx%d: %.0f pairs/s, kpts %d, valid %d" % (W, H, steps / dt, r.n_left, r.valid))
ctx.close()
