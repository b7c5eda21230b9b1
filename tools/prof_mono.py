import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
scene = synth.Scene(synth.SEEDS["C4"], W)
mono = [torch.from_numpy(synth.stereo_pair(scene, k, W, H)[0]).cuda() for k in (0, 4, 8)]
rig = synth.stereo_rig(W)
p = uvo.Params.mono(SURF_MIN_HESSIAN=6387, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8,
                    ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0, REPROJECTION_TOLERANCE=3.0)
ctx = uvo.Context(p, 0, W, H, 8192)
ctx.mono_set_camera(rig.K_left)
order = [0, 1, 2, 1]
for i in range(24):
    ctx.mono_step(mono[order[i % 4]], 4.0, 0.2)
ctx.close()
