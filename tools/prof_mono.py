"""Short C4 mono run for rocprofv3: python tools/prof_mono.py [contract|1px] [steps]
contract: the shipped thresholds 0.1 / 0.1 / 0.1 px with RANSAC for both estimators (BASELINE configs[3]); 1px: the variant of rounds 1-2.
Frames two steps apart (essential branch) and a quarter step apart (homography first), as tools/bench_configs.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
which = sys.argv[1] if len(sys.argv) > 1 else "contract"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene = synth.Scene(synth.SEEDS["C4"], W)
ks = [0, 2, 4, 2, 0, 0.25, 0.5, 0.25]
dev = {k: torch.from_numpy(synth.mono_frame(scene, k, W, H)).cuda() for k in sorted(set(ks))}
rig = synth.stereo_rig(W)
R0, C0 = synth.camera_pose(0)
rng = scene.depth_at_center(C0, R0)
kw = dict(SURF_MIN_HESSIAN=6456, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8)
if which == "1px":
    kw.update(ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0, REPROJECTION_TOLERANCE=3.0)
ctx = uvo.Context(uvo.Params.mono(**kw), 0, W, H, 8192)
ctx.mono_set_camera(rig.K_left)
nv = 0
for i in range(steps):
    r = ctx.mono_step(dev[ks[i % len(ks)]], rng, 0.2)
    nv += r.valid
print(which, "valid", nv, "of", steps, "matches", r.n_matches)
ctx.close()
