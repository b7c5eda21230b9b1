#!/usr/bin/env python3
"""Generates tests/golden/preproc_96x160.npz from the CPU oracle (oracle/o_preproc.c) on a seeded colour image.
SELF-GENERATED (the reference ships no data and OpenCV is not available): pins the oracle against drift and gives the
HIP path a file-based parity target.  Run:  python tests/golden/make_golden_preproc.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402


def main():
    rng = np.random.default_rng(20250911)
    H, W, DW = 96, 160, 100
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = np.clip((125 + 85 * np.sin(xx / 8.0) * np.cos(yy / 5.0))[..., None] + rng.normal(0, 14, (H, W, 3)) + np.array([8, -6, 15]),
                  0, 255).astype(np.uint8)
    dh = int(H / (W / DW))
    K = np.array([[92.0, 0, 51.0], [0, 94.0, 29.0], [0, 0, 1.0]])
    newK = np.array([[86.0, 0, 50.0], [0, 88.0, 30.0], [0, 0, 1.0]])
    dist = np.array([-0.22, 0.06, 0.0011, -0.0016])
    small = po.resize_area_c3(rgb, DW, dh)
    gray = po.rgb2gray(small)
    und = po.undistort(gray, K, dist, newK)
    eq = po.clahe(und, 8.0)
    full = po.get_image(rgb, DW, K, dist, newK, True, 8)
    assert np.array_equal(full, eq)
    np.savez_compressed(os.path.join(HERE, "preproc_96x160.npz"), rgb=rgb, K=K, newK=newK, dist=dist, desired_width=np.array([DW]),
                        clip_limit=np.array([8]), small=small, gray=gray, undistorted=und, out=eq)
    print("wrote preproc_96x160.npz", eq.shape)


if __name__ == "__main__":
    main()
