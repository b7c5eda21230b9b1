#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/) on seeded inputs.

The reference ships no tests, golden vectors or data and its arithmetic lives in OpenCV, which is
not available here, so these fixtures are SELF-GENERATED: they pin the oracle against drift and
give the HIP path a second, file-based parity target.  They are NOT outputs of the reference.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from ergo_uvo_amd import synth  # noqa: E402


def surf_image():
    sc = synth.Scene(777, 320)
    L, _ = synth.stereo_pair(sc, 0, 320, 180)
    return L


def main():
    # --- SURF on a 320x180 synthetic view ---
    img = surf_image()
    s = po.integral(img)
    det, tr = po.surf_layer(s, 15, 1)
    kps, desc = po.surf(img, 800)
    np.savez_compressed(os.path.join(HERE, "surf_320x180.npz"), img=img, integral_crc=np.array([int(s.astype(np.int64).sum())]),
                        integral_last=s[-1, -1:], det15_rows=det[60:64].copy(), trace15_rows=tr[60:64].copy(), kps=kps, desc=desc)
    # --- matcher: two descriptor sets (subset of the above, perturbed) ---
    rng = np.random.default_rng(42)
    d1 = desc[:120].copy()
    perm = rng.permutation(len(desc))[:150]
    d2 = (desc[perm] + rng.normal(0, 0.02, (150, 64))).astype(np.float32)
    idx, dist = po.knn2(d1, d2)
    m = po.match(d1, d2, 0.8)
    np.savez_compressed(os.path.join(HERE, "match_120x150.npz"), d1=d1, d2=d2, idx=idx, dist=dist, matches=m)
    # --- PnP RANSAC: 60 points, 25 % outliers ---
    rig = synth.stereo_rig(1280)
    n = 60
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    Rt, tt = synth.true_relative_motion()
    Y = X @ Rt.T + tt
    K = rig.K_left
    x = (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, 0.3, (n, 2))
    bad = rng.random(n) < 0.25
    x[bad] = rng.uniform(0, 700, (int(bad.sum()), 2))
    x = x.astype(np.float32)
    ok, rvec, tvec, inl = po.solve_pnp_ransac(X, x, K, 1000, 1.0, 0.99)
    np.savez_compressed(os.path.join(HERE, "pnp_60.npz"), X=X, x=x, K=K, ok=np.array([ok]), rvec=rvec, tvec=tvec, inliers=inl)
    # --- triangulation + extract_3Dpoints: 40 stereo correspondences ---
    n = 40
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    P1 = rig.K_left @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = rig.K_right @ np.hstack([rig.R_right, rig.t_right[:, None]])

    def proj(P, X):
        Y = np.hstack([X, np.ones((len(X), 1))]) @ P.T
        return Y[:, :2] / Y[:, 2:]
    x1 = (proj(P1, X) + rng.normal(0, 0.4, (n, 2))).astype(np.float32)
    x2 = (proj(P2, X) + rng.normal(0, 0.4, (n, 2))).astype(np.float32)
    x2[::9] += 25.0                                  # gross mismatches -> filtered by the reprojection test
    p4 = po.triangulate(P1, P2, x1, x2)
    pts, idx3 = po.extract_3d_points(x1, x2, np.eye(3), np.zeros(3), rig.R_right, rig.t_right, rig.K_left, rig.K_right, p4)
    np.savez_compressed(os.path.join(HERE, "tri_40.npz"), P1=P1, P2=P2, x1=x1, x2=x2, points4d=p4, pts=pts, idx=idx3,
                        K1=rig.K_left, K2=rig.K_right, R2=rig.R_right, t2=rig.t_right)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
