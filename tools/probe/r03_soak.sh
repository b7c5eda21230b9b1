# Long runs: 20000 pairs of the headline pipeline, 3000 pairs of the stereo loop on SIFT at 1080p (every pose must be valid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20000 --warmup 20 --timed-only 2>/dev/null | tail -1
python3 - <<'PY'
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(a).cuda() for a in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(), 0, W, H, 12288)
ctx.set_feature_detector("SIFT")
ctx.stereo_set_depth(4)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
n = 3000
ctx.stereo_submit(*dev[0]); ctx.stereo_collect(0.05)
t0 = time.perf_counter(); sub = 1; nv = 0; inl = []
for i in range(1, n):
    while sub < n and sub - i < 4:
        ctx.stereo_submit(*dev[order[sub % 6]]); sub += 1
    r = ctx.stereo_collect(0.05); nv += r.valid; inl.append(r.n_inliers)
dt = time.perf_counter() - t0
print("SIFT stereo loop soak: %d pairs, %d valid, %.1f pairs/s, inliers min %d max %d" % (n - 1, nv, (n - 1) / dt, min(inl), max(inl)))
ctx.close()
PY
