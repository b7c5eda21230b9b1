"""Deterministic synthetic stereo / mono sequences for tests and bench.py.

The reference ships no data (its only data set, ``uvo/bags/test.bag``, must be downloaded:
README.md:78-80, bags/README.md:1), so workloads are rendered here from a seeded scene
(SURVEY.md 8(d)): a smooth non-planar surface at ~4 m textured with Gaussian blobs over a
wide range of scales, seen through the shipped stereo rig (stereo_VO_intrinsics.yaml:7-53,
intrinsics scaled to the target width, zero distortion, 0.33 m baseline), moving with a fixed
small SE(3) step per frame.  NumPy float64 only, so the same seed gives the same bytes here
and on the GPU box (``scene_digest`` lets a test pin that).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass

import numpy as np
from scipy import ndimage

# stereo_VO_intrinsics.yaml:7-17 (1280-wide calibration) and :45-53 (extrinsics)
_FX_L, _FY_L, _CX_L, _CY_L = 1.335036735254999e+03, 1.332419247540885e+03, 0.644564474737301e+03, 0.357685235527149e+03
_FX_R, _FY_R, _CX_R, _CY_R = 1.330461901943011e+03, 1.328225165048530e+03, 0.684598875987595e+03, 0.382841174819059e+03
_T_RIGHT = np.array([-0.33, 0.0, 0.0])
_CAL_WIDTH = 1280.0

SEEDS = {"C1": 20250904, "C2": 20250905, "C3": 20250906, "C4": 20250907, "C5": 20250910}


@dataclass
class Rig:
    K_left: np.ndarray
    K_right: np.ndarray
    R_right: np.ndarray
    t_right: np.ndarray


def stereo_rig(width: int) -> Rig:
    s = width / _CAL_WIDTH
    KL = np.array([[_FX_L * s, 0, _CX_L * s], [0, _FY_L * s, _CY_L * s], [0, 0, 1.0]])
    KR = np.array([[_FX_R * s, 0, _CX_R * s], [0, _FY_R * s, _CY_R * s], [0, 0, 1.0]])
    return Rig(KL, KR, np.eye(3), _T_RIGHT.copy())


def _rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


# per-frame motion of the (left) camera, expressed in the previous camera frame
STEP_T = np.array([0.03, -0.01, 0.05])
STEP_R = _rot_xyz(np.deg2rad(0.4), np.deg2rad(-0.3), np.deg2rad(0.2))


class Scene:
    """Heightfield z = Z(X, Y) over the world XY plane, textured in world coordinates."""

    TEXEL = 0.002  # metres per texel at the nominal 1920-wide resolution

    def __init__(self, seed: int, width: int = 1920, extent=(-4.6, 4.6, -3.2, 3.2)):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.texel = self.TEXEL * (1920.0 / width)
        self.x0, self.x1, self.y0, self.y1 = extent
        # depth field: 4 m + six low-frequency cosines, total amplitude <= 1.5 m
        self.amp = rng.uniform(0.10, 0.25, 6)
        lam = rng.uniform(2.5, 6.0, 6)
        ang = rng.uniform(0, 2 * np.pi, 6)
        self.kx = 2 * np.pi / lam * np.cos(ang)
        self.ky = 2 * np.pi / lam * np.sin(ang)
        self.ph = rng.uniform(0, 2 * np.pi, 6)
        tw = int(round((self.x1 - self.x0) / self.texel))
        th = int(round((self.y1 - self.y0) / self.texel))
        tex = np.zeros((th, tw), np.float64)
        # blobs: impulses of random sign/amplitude filtered at 8 log-spaced sigmas in [1.2, 14] texels
        sigmas = np.exp(np.linspace(np.log(1.2), np.log(14.0), 8))
        n_total = int(40000 * (tw * th) / (1920.0 * 1080.0))
        for sg in sigmas:
            n = n_total // len(sigmas)
            imp = np.zeros((th, tw), np.float64)
            ys = rng.integers(0, th, n)
            xs = rng.integers(0, tw, n)
            a = rng.uniform(15.0, 90.0, n) * rng.choice([-1.0, 1.0], n)
            np.add.at(imp, (ys, xs), a * (2 * np.pi * sg * sg))  # unit-peak Gaussians of amplitude a
            tex += ndimage.gaussian_filter(imp, sg, mode="constant", truncate=4.0)
        self.tex = (128.0 + tex).astype(np.float32)
        self.noise_seed = seed ^ 0x5EED

    def Z(self, X, Y):
        z = np.full_like(X, 4.0)
        for a, kx, ky, ph in zip(self.amp, self.kx, self.ky, self.ph):
            z += a * np.cos(kx * X + ky * Y + ph)
        return z

    def render(self, K: np.ndarray, R_wc: np.ndarray, C_w: np.ndarray, width: int, height: int, noise_key: int) -> np.ndarray:
        """Image of the scene from a camera with centre C_w and rotation R_wc (world->camera)."""
        u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
        d_cam = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
        d_w = d_cam @ R_wc  # R_wc^T applied to row vectors
        t = np.full_like(u, 4.0)
        for _ in range(10):  # fixed-point ray / heightfield intersection
            X = C_w[0] + t * d_w[..., 0]
            Y = C_w[1] + t * d_w[..., 1]
            t = (self.Z(X, Y) - C_w[2]) / d_w[..., 2]
        X = C_w[0] + t * d_w[..., 0]
        Y = C_w[1] + t * d_w[..., 1]
        tx = (X - self.x0) / self.texel - 0.5
        ty = (Y - self.y0) / self.texel - 0.5
        img = ndimage.map_coordinates(self.tex, [ty, tx], order=1, mode="nearest")
        nrng = np.random.default_rng((self.noise_seed, noise_key))
        img = img + nrng.integers(-2, 3, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def depth_at_center(self, C_w, R_wc):
        d = R_wc.T @ np.array([0, 0, 1.0])
        t = 4.0
        for _ in range(10):
            t = (self.Z(np.array(C_w[0] + t * d[0]), np.array(C_w[1] + t * d[1])) - C_w[2]) / d[2]
        return float(t)


def _frac_step(f: float):
    """A fraction f of the per-frame step: rotation about STEP_R's axis by f times its angle, translation f * STEP_T."""
    from scipy.spatial.transform import Rotation
    rv = Rotation.from_matrix(STEP_R).as_rotvec()
    return Rotation.from_rotvec(f * rv).as_matrix(), f * STEP_T


def camera_pose(k):
    """World->camera rotation and camera centre of the left camera at frame k (world = frame 0).  k may be negative
    (the motion run backwards) or fractional (whole steps, then a fraction of one step: low-parallax frames)."""
    R = np.eye(3)
    C = np.zeros(3)
    whole = int(np.floor(k)) if k >= 0 else -int(np.floor(-k))
    for _ in range(max(whole, 0)):
        # X_new = STEP_R (X_old - STEP_T)
        C = C + R.T @ STEP_T
        R = STEP_R @ R
    for _ in range(max(-whole, 0)):
        # the inverse step: X_old = STEP_R^T X_new + STEP_T
        R = STEP_R.T @ R
        C = C - R.T @ STEP_T
    f = float(k) - whole
    if f != 0.0:
        Rf, Tf = _frac_step(f)
        C = C + R.T @ Tf
        R = Rf @ R
    return R, C


def mono_frame(scene: Scene, k, width: int, height: int):
    """Left view at (possibly negative or fractional) frame k; the noise key is derived from k so frames differ."""
    rig = stereo_rig(width)
    R, C = camera_pose(k)
    return scene.render(rig.K_left, R, C, width, height, int(round(k * 64)) * 2 + (1 << 20))


def stereo_pair(scene: Scene, k: int, width: int, height: int):
    rig = stereo_rig(width)
    R, C = camera_pose(k)
    left = scene.render(rig.K_left, R, C, width, height, 2 * k)
    # X_right = R_right X_left + t_right  =>  centre of the right camera in left coords = -R_right^T t_right
    C_r = C + R.T @ (-rig.R_right.T @ rig.t_right)
    right = scene.render(rig.K_right, rig.R_right @ R, C_r, width, height, 2 * k + 1)
    return left, right


def stereo_sequence(seed: int, width: int, height: int, n_frames: int):
    scene = Scene(seed, width)
    return [stereo_pair(scene, k, width, height) for k in range(n_frames)]


def true_relative_motion():
    """(R, t) with X_curr = R X_prev + t for consecutive frames."""
    return STEP_R.copy(), -STEP_R @ STEP_T


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
