import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
scene = synth.Scene(synth.SEEDS["C4"], W)
ks = [0, 2, 4, 2, 0, 0.25, 0.5, 0.25]
dev = {k: torch.from_numpy(synth.mono_frame(scene, k, W, H)).cuda() for k in sorted(set(ks))}
rig = synth.stereo_rig(W)
R0, C0 = synth.camera_pose(0)
rng = scene.depth_at_center(C0, R0)
ctx = uvo.Context(uvo.Params.mono(SURF_MIN_HESSIAN=6456, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8), 0, W, H, 8192)
ctx.mono_set_camera(rig.K_left)
for i in range(8): ctx.mono_step(dev[ks[i % 8]], rng, 0.2)
T = {i: [] for i in range(8)}
info = {}
for rep in range(12):
    for i in range(8):
        torch.cuda.synchronize(); a = time.perf_counter()
        r = ctx.mono_step(dev[ks[i]], rng, 0.2)
        T[i].append((time.perf_counter() - a) * 1e3)
        info[i] = (r.used_essential, r.success, r.n_matches, r.n_inliers, r.n_good3d)
for i in range(8):
    print("frame %d (k=%s <- %s): %.3f ms  essential %d success %d M %d inl %d G %d" % (i, ks[i], ks[i - 1], np.median(T[i]), *info[i]))
ctx.close()
