#!/usr/bin/env python3
"""Generates tests/golden/jpeg_cases.npz: JPEG byte streams and the pixels a REAL libjpeg-turbo decodes from them
(Pillow bundles libjpeg-turbo; its decoder runs libjpeg's defaults -- JDCT_ISLOW, fancy upsampling -- which are also
cv::imdecode's).  This is the one part of the oracle that is pinned to a third-party implementation:
    python tests/golden/make_golden_codec.py
The fixture holds data only (compressed bytes in, decoded bytes out)."""
import io
import os

import numpy as np
from PIL import Image, features
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(rng, h, w, c):
    a = rng.normal(size=(h, w, c)) * 60
    a = ndimage.gaussian_filter(a, (3, 3, 0)) * 6 + 128 + rng.normal(size=(h, w, c)) * 8
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    assert features.check("libjpeg_turbo"), "Pillow without libjpeg-turbo"
    rng = np.random.default_rng(20250911)
    cases = [("c420_q75", 64, 48, 3, dict(quality=75, subsampling=2)), ("c422_q90", 37, 53, 3, dict(quality=90, subsampling=1)),
             ("c444_q95", 40, 40, 3, dict(quality=95, subsampling=0)), ("grey_q80", 33, 71, 1, dict(quality=80)),
             ("c420_odd_q60", 121, 97, 3, dict(quality=60, subsampling=2)), ("c420_rst", 100, 130, 3, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
             ("c420_opt", 72, 88, 3, dict(quality=85, subsampling=2, optimize=True)), ("tiny", 1, 1, 3, dict(quality=75, subsampling=2)),
             ("c422_wide", 7, 200, 3, dict(quality=70, subsampling=1)), ("c420_q100", 50, 50, 3, dict(quality=100, subsampling=2))]
    out = {"names": np.array([c[0] for c in cases]), "libjpeg_turbo": np.frombuffer(features.version("jpg").encode(), np.uint8)}
    for name, h, w, c, kw in cases:
        img = picture(rng, h, w, c)
        b = io.BytesIO()
        Image.fromarray(img[..., 0] if c == 1 else img).save(b, "JPEG", **kw)
        data = b.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(data)))
        out[name + "_jpeg"] = np.frombuffer(data, np.uint8)
        out[name + "_rgb"] = dec                      # RGB order (cv::imdecode returns the same bytes as BGR)
    np.savez(os.path.join(HERE, "jpeg_cases.npz"), **out)
    print("wrote jpeg_cases.npz:", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")), "bytes")


if __name__ == "__main__":
    main()
