"""Where the submitting thread's time goes in the pipelined C3 loop: wall time inside stereo_submit vs stereo_collect (a collect that
returns at once means the host, not the device, paces the pipeline).  python tools/probe/host_split.py [depth] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
ctx.stereo_set_depth(depth)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
for phase in ("warm", "timed"):
    n = 60 if phase == "warm" else steps
    t_sub = t_col = 0.0
    sub = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        while sub < n and sub - i < depth:
            a = time.perf_counter(); ctx.stereo_submit(*dev[order[sub % 6]]); t_sub += time.perf_counter() - a; sub += 1
        a = time.perf_counter(); ctx.stereo_collect(0.05); t_col += time.perf_counter() - a
    tot = time.perf_counter() - t0
    if phase == "timed":
        print("depth %d: %.0f pairs/s; per pair: submit %.1f us, collect %.1f us, rest %.1f us" % (depth, n / tot, t_sub / n * 1e6, t_col / n * 1e6, (tot - t_sub - t_col) / n * 1e6))
ctx.close()
