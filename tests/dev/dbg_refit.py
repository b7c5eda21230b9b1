"""Refit diagnostics on the GPU box: HIP solvePnPRansac vs the oracle for growing inlier counts (relative pose differences),
then a C3 run with UVO_DBG_PHASE for the refit's phase times.  python tests/dev/dbg_refit.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
from oracle import pyoracle as po

rig = synth.stereo_rig(1920)
ctx = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
K = rig.K_left
Rt, tt = synth.true_relative_motion()
for n, outl, noise in [(30, 0.0, 0.2), (100, 0.2, 0.3), (500, 0.3, 0.5), (2500, 0.25, 0.4), (6000, 0.5, 0.7), (2000, 0.0, 0.0)]:
    rng = np.random.default_rng(n)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    Y = X @ Rt.T + tt
    x = (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outl
    x[bad] = rng.uniform(0, 1000, (int(bad.sum()), 2))
    x = x.astype(np.float32)
    ok, rv, tv, inl = ctx.solvePnPRansac(X, x, K, 1000, 1.0, 0.99)
    ook, orv, otv, oinl = po.solve_pnp_ransac(X, x, K, 1000, 1.0, 0.99)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
    print(f"n={n} inliers={len(inl)} same_inliers={np.array_equal(inl, oinl)} rel_rvec={rel(rv, orv):.3e} rel_tvec={rel(tv, otv):.3e} tvec={tv}")
ctx.close()
