#!/usr/bin/env python3
"""Deterministic inputs for the optional OpenCV dumper (tools/opencv_oracle/opencv_oracle.cpp) and for
tests/test_opencv_fixture.py, which regenerates them to feed the oracle / the HIP path with exactly what OpenCV saw.
    python tools/opencv_oracle/make_inputs.py /tmp/uvo_inputs.npz [--width 640]
numpy / scipy only (ergo_uvo_amd.synth); the same seed gives the same bytes everywhere."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _stereo_points(n, seed, rig, noise=0.3):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)

    def proj(K, R, t, X):
        Y = X @ R.T + t
        return (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    x1 = proj(rig.K_left, np.eye(3), np.zeros(3), X) + rng.normal(0, noise, (n, 2))
    x2 = proj(rig.K_right, rig.R_right, rig.t_right, X) + rng.normal(0, noise, (n, 2))
    P1 = rig.K_left @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = rig.K_right @ np.hstack([rig.R_right, rig.t_right[:, None]])
    return X, x1.astype(np.float32), x2.astype(np.float32), P1, P2


def _mono_points(n, seed, planar, noise, outliers):
    from scipy.spatial.transform import Rotation
    K = np.array([[800.0, 0, 320.0], [0, 790.0, 240.0], [0, 0, 1.0]])
    rng = np.random.default_rng(seed)
    R = Rotation.from_rotvec([0.02, -0.03, 0.015]).as_matrix()
    t = np.array([0.30, -0.08, 0.12])
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(3.0, 7.0, n)], 1)
    if planar:
        nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
        X[:, 2] = (5.0 - X[:, 0] * nrm[0] - X[:, 1] * nrm[1]) / nrm[2]

    def proj(Y):
        return (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    x1 = proj(X) + rng.normal(0, noise, (n, 2))
    x2 = proj(X @ R.T + t) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    x2[bad] = rng.uniform(0, 600, (int(bad.sum()), 2))
    return K, x1.astype(np.float32), x2.astype(np.float32)


def build_inputs(width=640, seed=123, min_hessian=1500):
    from ergo_uvo_amd import synth
    height = width * 9 // 16
    scene = synth.Scene(seed, width)
    (l0, r0), (l1, _r1) = synth.stereo_pair(scene, 0, width, height), synth.stereo_pair(scene, 1, width, height)
    rig = synth.stereo_rig(width)
    d = dict(left0=l0, right0=r0, left1=l1, min_hessian=np.array(min_hessian, np.int32), lowe_ratio=np.array(0.8),
             K_left=rig.K_left, K_right=rig.K_right, R_right=rig.R_right, t_right=rig.t_right)
    X, x1, x2, P1, P2 = _stereo_points(500, 11, rig)
    d.update(tri_P1=P1, tri_P2=P2, tri_x1=x1, tri_x2=x2)
    # PnP: 3-D points in the previous camera, pixels in the current one, 25 % outliers
    rng = np.random.default_rng(17)
    n = 800
    Xp = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    Rt, tt = synth.true_relative_motion()
    Y = Xp @ Rt.T + tt
    K = rig.K_left
    xp = (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, 0.4, (n, 2))
    bad = rng.random(n) < 0.25
    xp[bad] = rng.uniform(0, width, (int(bad.sum()), 2))
    d.update(pnp_X=Xp, pnp_x=xp.astype(np.float32), pnp_K=K, pnp_iterations=np.array(1000, np.int32), pnp_reprojection_error=np.array(1.0),
             pnp_confidence=np.array(0.99))
    mK, e1, e2 = _mono_points(600, 101, False, 0.15, 0.3)
    _, h1, h2 = _mono_points(600, 202, True, 0.10, 0.3)
    d.update(mono_K=mK, e_x1=e1, e_x2=e2, h_x1=h1, h_x2=h2, ransac_threshold=np.array(1.0), lmeds_threshold=np.array(0.1),
             e_confidence=np.array(0.99), h_confidence=np.array(0.99), max_iters=np.array(2000, np.int32))
    # get_image: a colour frame twice the target width, mild distortion
    rng = np.random.default_rng(9)
    H2, W2, DW = 360, 640, 320
    yy, xx = np.mgrid[0:H2, 0:W2]
    rgb = np.clip((120 + 90 * np.sin(xx / 11.0) * np.cos(yy / 6.0))[..., None] + rng.normal(0, 10, (H2, W2, 3)), 0, 255).astype(np.uint8)
    pK = np.array([[260.0, 0, 161.0], [0, 262.0, 88.0], [0, 0, 1.0]])
    pnew = np.array([[250.0, 0, 160.0], [0, 252.0, 89.5], [0, 0, 1.0]])
    d.update(pre_rgb=rgb, pre_K=pK, pre_dist=np.array([-0.25, 0.07, 0.001, -0.002]), pre_newK=pnew, pre_width=np.array(DW, np.int32),
             pre_clip_limit=np.array(3.0))
    # resize_camera_matrix: the shipped stereo calibration (left camera of stereo_VO_intrinsics.yaml:7-25: values, not file text)
    d.update(cam_K=np.array([[1335.036735254999, 0, 644.564474737301], [0, 1332.419247540885, 357.685235527149], [0, 0, 1.0]]),
             cam_dist=np.array([0.475667186716851, 0.126480045385593, 0.0, 0.0]), cam_width=np.array(1280, np.int32), cam_height=np.array(720, np.int32),
             cam_desired_width=np.array(640, np.int32))
    return d


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "uvo_inputs.npz"
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 640
    np.savez(out, **build_inputs(width))
    print("wrote", out)
