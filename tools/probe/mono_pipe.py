"""Pipelined mono throughput (uvo_mono_submit / uvo_mono_collect) at C4: python tools/probe/mono_pipe.py [depth]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 14
scene = synth.Scene(synth.SEEDS["C4"], W)
dmono = [torch.from_numpy(synth.stereo_pair(scene, k, W, H)[0]).cuda() for k in (0, 4, 8, 12)]
rig = synth.stereo_rig(W)
R0, C0 = synth.camera_pose(0)
rng = scene.depth_at_center(C0, R0)
p = uvo.Params.mono(SURF_MIN_HESSIAN=6387, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8, ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0,
                    REPROJECTION_TOLERANCE=3.0)
ctx = uvo.Context(p, 0, W, H, 8192)
ctx.mono_set_camera(rig.K_left)
ctx.stereo_set_depth(depth)
order = [0, 1, 2, 3, 2, 1]
for steps in (24, 600):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sub = nv = 0
    for i in range(steps):
        while sub < steps and sub - i < depth:
            ctx.mono_submit(dmono[order[sub % 6]], rng); sub += 1
        nv += ctx.mono_collect(0.2).valid
    dt = time.perf_counter() - t0
print("depth %d max_b_mono %s a_overlap_mono %s: %.1f frames/s, valid %d" % (depth, os.environ.get("UVO_MAX_B_MONO", "10"), os.environ.get("UVO_A_OVERLAP_MONO", "4"), steps / dt, nv))
ctx.close()
