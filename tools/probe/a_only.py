"""Pipelined throughput with and without the PnP stage (MIN_NUM_3DPOINTS raised so VO:634 skips it): how much do the
stage-B kernels cost the stage-A kernels they run beside?   python tools/probe/a_only.py [steps]"""
import os, sys, time, itertools
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
for label, kw in [("A+B", {}), ("A only", {"MIN_NUM_3DPOINTS": 1000000}), ("1 hyp", {"ITERATIONS_COUNT": 1}), ("A+B", {}), ("1 hyp", {"ITERATIONS_COUNT": 1}), ("64 hyp", {"ITERATIONS_COUNT": 64})]:
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387, **kw), 0, W, H, 8192)
    ctx.stereo_set_depth(6)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    order = itertools.cycle([0, 1, 2, 3, 2, 1])
    for _ in range(20):
        ctx.stereo_step(*dev[next(order)], 0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sub = 0
    for i in range(steps):
        while sub < steps and sub - i < 6:
            ctx.stereo_submit(*dev[next(order)]); sub += 1
        r = ctx.stereo_collect(0.05)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-7s %8.1f pairs/s  (%.1f us/pair)  last: valid %d good3d %d inl %d" % (label, steps / dt, dt / steps * 1e6, r.valid, r.n_good3d, r.n_inliers))
    ctx.close()
