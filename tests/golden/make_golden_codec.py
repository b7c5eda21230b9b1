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


def inflate_quant(data: bytes, factor: int) -> bytes:
    """The same entropy-coded data with every 8-bit quantisation step multiplied by `factor` (capped at 255): the dequantised
    coefficients overshoot, so the IDCT output leaves [-128, 383] and libjpeg's post-IDCT range_limit table is read in its
    saturated and wrapped parts (jdmaster.c prepare_range_limit_table) -- what a damaged or hostile stream does."""
    b = bytearray(data)
    i = 2
    while i + 4 <= len(b) and b[i] == 0xFF:
        m, ln = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xDB:
            j = i + 4
            while j < i + 2 + ln:
                assert b[j] >> 4 == 0, "8-bit tables only"
                for k in range(j + 1, j + 65):
                    b[k] = min(255, b[k] * factor)
                j += 65
        if m == 0xDA:
            break
        i += 2 + ln
    return bytes(b)


def decode_scalar(data: bytes):
    """libjpeg-turbo's decode with its SIMD extensions off (JSIMD_FORCENONE=1, read when the library initialises, hence a child
    process): jidctint.c + the range_limit table, the algorithm the oracle and the HIP kernel restate.  On in-range streams the
    SIMD IDCT gives the same bytes (the ten ordinary cases above are decoded with SIMD on); on these overshooting streams it
    does not -- it dequantises with 16-bit wrap-around and saturates instead of reading the table -- so "what libjpeg decodes"
    from a hostile stream depends on the CPU it runs on, and the scalar path is the one pinned here."""
    import subprocess
    import sys
    code = ("import io,sys,numpy as np\nfrom PIL import Image\n"
            "a=np.asarray(Image.open(io.BytesIO(sys.stdin.buffer.read())))\n"
            "sys.stdout.buffer.write(np.array(a.shape+(0,)*(3-a.ndim),np.int32).tobytes()+a.tobytes())")
    r = subprocess.run([sys.executable, "-c", code], input=data, stdout=subprocess.PIPE, check=True, env=dict(os.environ, JSIMD_FORCENONE="1"))
    shp = [int(v) for v in np.frombuffer(r.stdout[:12], np.int32) if v]
    return np.frombuffer(r.stdout[12:], np.uint8).reshape(shp).copy()


def harsh_picture(rng, h, w, c):
    a = rng.integers(0, 2, (h // 4 + 1, w // 4 + 1, c)) * 255
    a = np.kron(a, np.ones((4, 4, 1)))[:h, :w]
    a = a + rng.normal(size=(h, w, c)) * 30
    return np.clip(a, 0, 255).astype(np.uint8)


def make_png(rng):
    """tests/golden/png_cases.npz: PNG streams written by Pillow and the pixels PILLOW's own decoder (zlib) reads back from them,
    in cv::imdecode(IMREAD_UNCHANGED)'s channel order -- PNG is lossless and its decoding is fixed by the specification, so this
    pins the oracle's and the product's inflate / filter / expansion code to an independent implementation."""
    cases = [("grey8", 33, 71, "L", {}), ("rgb8", 64, 48, "RGB", {}), ("rgb8_l9", 121, 97, "RGB", dict(compress_level=9, optimize=True)),
             ("rgb8_stored", 20, 30, "RGB", dict(compress_level=0)), ("rgba8", 40, 56, "RGBA", {}), ("pal8", 50, 50, "P64", {}),
             ("pal4", 37, 53, "P11", {}), ("bilevel", 17, 70, "1", {}), ("grey4", 24, 31, "L", dict(bits=4)), ("tiny", 1, 1, "RGB", {}),
             ("photo", 120, 160, "PHOTO", dict(compress_level=6))]
    out = {"names": np.array([c[0] for c in cases])}
    for name, h, w, mode, kw in cases:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if mode == "PHOTO":
            a = np.dstack([picture(rng, h, w, 3), a[..., 3:]]); mode = "RGB"           # smooth content: exercises the Paeth / Average filters
        img = {"L": lambda: Image.fromarray(a[..., 0]), "RGB": lambda: Image.fromarray(a[..., :3]), "RGBA": lambda: Image.fromarray(a),
               "P64": lambda: Image.fromarray(a[..., :3]).quantize(64), "P11": lambda: Image.fromarray(a[..., :3]).quantize(11),
               "1": lambda: Image.fromarray(a[..., 0] > 127)}[mode]()
        b = io.BytesIO()
        img.save(b, "PNG", **kw)
        data = b.getvalue()
        ref = Image.open(io.BytesIO(data))
        if ref.mode == "P":
            px = np.asarray(ref.convert("RGB"))[..., ::-1]
        elif ref.mode in ("L", "1"):
            px = np.asarray(ref.convert("L"))
        elif ref.mode == "RGB":
            px = np.asarray(ref)[..., ::-1]
        else:
            px = np.asarray(ref)[..., [2, 1, 0, 3]]
        out[name + "_png"] = np.frombuffer(data, np.uint8)
        out[name + "_px"] = np.ascontiguousarray(px)
    np.savez(os.path.join(HERE, "png_cases.npz"), **out)
    print("wrote png_cases.npz:", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")), "bytes")


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
    # streams whose IDCT output overshoots far beyond a sample's range (v in [384, 511] must read 255, not 0)
    sat = [("sat_c420_x6", 40, 56, 3, dict(quality=50, subsampling=2), 6), ("sat_c444_x12", 24, 24, 3, dict(quality=60, subsampling=0), 12),
           ("sat_grey_x9", 32, 40, 1, dict(quality=40), 9), ("sat_c422_x255", 16, 32, 3, dict(quality=90, subsampling=1), 255)]
    for name, h, w, c, kw, factor in sat:
        img = harsh_picture(rng, h, w, c)
        b = io.BytesIO()
        Image.fromarray(img[..., 0] if c == 1 else img).save(b, "JPEG", **kw)
        data = inflate_quant(b.getvalue(), factor)
        out[name + "_jpeg"] = np.frombuffer(data, np.uint8)
        out[name + "_rgb"] = decode_scalar(data)
    out["names"] = np.array([c[0] for c in cases] + [c[0] for c in sat])
    np.savez(os.path.join(HERE, "jpeg_cases.npz"), **out)
    make_png(rng)
    print("wrote jpeg_cases.npz:", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")), "bytes")


if __name__ == "__main__":
    main()
