"""Short C3 stereo run for rocprofv3 / UVO_DBG_PHASE diagnostics: python tools/prof_stereo.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
for i in range(steps):
    r = ctx.stereo_step(*dev[order[i % 6]], 0.05)
print("valid", r.valid, "inliers", r.n_inliers, "kpts", r.n_left)
ctx.close()
