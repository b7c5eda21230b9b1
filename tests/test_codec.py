"""Compressed-image ingest: from_ros_to_cv_image = cv::imdecode (libjpeg) + COLOR_BayerBGGR2BGR
(uvo_libraries/src/math_utility.cpp:154-173).

The JPEG decode is the one component whose oracle is PINNED to a real third-party implementation: tests/golden/jpeg_cases.npz
holds JPEG streams and the pixels libjpeg-turbo (Pillow's) decoded from them; the oracle and the HIP path must reproduce them
byte for byte.  When Pillow is importable the CPU test also cross-checks a sweep of sizes / samplings / qualities live.
Bayer demosaicing has no third-party witness here (unpinned, recalled OpenCV behaviour): HIP <-> oracle + analytic cases.
PNG payloads (round 3) are lossless and their decoding is fixed by the specification: tests/golden/png_cases.npz holds streams and
the pixels Pillow's own decoder (zlib) reads from them -- a second third-party pin; the oracle, the host half of the product's
decoder (compiled for the CPU) and the HIP path must reproduce them byte for byte."""
import io
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cases():
    return np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))


def _bgr(rgb):
    return rgb if rgb.ndim == 2 else np.ascontiguousarray(rgb[..., ::-1])


def test_oracle_jpeg_equals_libjpeg_turbo_fixture(cases, oracle):
    for name in cases["names"]:
        got = oracle.jpeg_decode(bytes(cases[f"{name}_jpeg"]))
        assert np.array_equal(got, _bgr(cases[f"{name}_rgb"])), name


def test_oracle_jpeg_equals_libjpeg_turbo_live(oracle):
    PIL = pytest.importorskip("PIL.Image")
    from scipy import ndimage
    rng = np.random.default_rng(3)
    n = 0
    for (h, w) in [(8, 8), (16, 16), (37, 53), (1, 9), (123, 77), (240, 320)]:
        for ss in (0, 1, 2):
            for q in (40, 85, 100):
                a = ndimage.gaussian_filter(rng.normal(size=(h, w, 3)) * 60, (2, 2, 0)) * 5 + 128 + rng.normal(size=(h, w, 3)) * 6
                img = np.clip(a, 0, 255).astype(np.uint8)
                b = io.BytesIO()
                PIL.fromarray(img).save(b, "JPEG", quality=q, subsampling=ss)
                ref = np.asarray(PIL.open(io.BytesIO(b.getvalue())))
                assert np.array_equal(oracle.jpeg_decode(b.getvalue()), _bgr(ref)), (h, w, ss, q)
                n += 1
    assert n == 54


@pytest.fixture(scope="module")
def png_cases():
    return np.load(os.path.join(ROOT, "tests", "golden", "png_cases.npz"))


def _png_sweep():
    """(tag, bytes, expected pixels in imdecode's channel order) for a sweep of sizes, kinds and compression levels, via Pillow."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(8)
    for (h, w) in [(1, 1), (7, 13), (121, 97), (240, 320)]:
        for mode in ("L", "RGB", "RGBA", "P64", "P11", "1", "L4"):
            for kw in ({}, dict(compress_level=0), dict(compress_level=9, optimize=True)):
                a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
                if kw:
                    a[..., :3] = (a[..., :3] // 32) * 32
                img = {"L": lambda: PIL.fromarray(a[..., 0]), "L4": lambda: PIL.fromarray(a[..., 0]), "RGB": lambda: PIL.fromarray(a[..., :3]),
                       "RGBA": lambda: PIL.fromarray(a), "P64": lambda: PIL.fromarray(a[..., :3]).quantize(64),
                       "P11": lambda: PIL.fromarray(a[..., :3]).quantize(11), "1": lambda: PIL.fromarray(a[..., 0] > 127)}[mode]()
                b = io.BytesIO()
                img.save(b, "PNG", **(dict(kw, bits=4) if mode == "L4" else kw))
                ref = PIL.open(io.BytesIO(b.getvalue()))
                if ref.mode == "P":
                    px = np.asarray(ref.convert("RGB"))[..., ::-1]
                elif ref.mode in ("L", "1"):
                    px = np.asarray(ref.convert("L"))
                elif ref.mode == "RGB":
                    px = np.asarray(ref)[..., ::-1]
                else:
                    px = np.asarray(ref)[..., [2, 1, 0, 3]]
                yield (h, w, mode, tuple(kw.items())), b.getvalue(), np.ascontiguousarray(px)


def test_oracle_png_equals_pillow_fixture_and_live(png_cases, oracle):
    for name in png_cases["names"]:
        got = oracle.png_decode(bytes(png_cases[f"{name}_png"]))
        assert got.shape == png_cases[f"{name}_px"].shape and np.array_equal(got, png_cases[f"{name}_px"]), name
    n = 0
    for tag, data, px in _png_sweep():
        got = oracle.png_decode(data)
        assert got.shape == px.shape and np.array_equal(got, px), tag
        n += 1
    assert n == 84


def _png_with(data: bytes, **ihdr):
    """the same PNG with IHDR fields replaced (and the chunk's CRC recomputed)"""
    import struct
    import zlib
    b = bytearray(data)
    off = {"depth": 24, "ctype": 25, "interlace": 28}
    for k, v in ihdr.items():
        b[off[k]] = v
    b[29:33] = struct.pack(">I", zlib.crc32(bytes(b[12:29])))
    return bytes(b)


def test_oracle_png_refusals(png_cases, oracle):
    rgb = bytes(png_cases["rgb8_png"])
    for bad in (_png_with(rgb, interlace=1), _png_with(rgb, depth=16), _png_with(rgb, ctype=4), rgb[:len(rgb) // 2],
                rgb[:60] + bytes([rgb[60] ^ 0x40]) + rgb[61:]):
        with pytest.raises(ValueError):
            oracle.png_decode(bad)


def test_product_png_host_half_on_cpu(png_cases, tmp_path):
    """ergo_uvo_amd/csrc/uvo_png.h (container, inflate, filters) is plain C++: compiled with g++ and run here, its unfiltered
    samples must be the ones Pillow reads (8-bit kinds compared sample for sample; sub-byte kinds after unpacking)."""
    import shutil
    import struct
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "png_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "png_host.cpp")])
    PIL = pytest.importorskip("PIL.Image")

    def run(data):
        (tmp_path / "a.png").write_bytes(data)
        r = subprocess.run([exe, str(tmp_path / "a.png"), str(tmp_path / "a.bin")], capture_output=True, text=True)
        if r.returncode != 0:
            return None, r.stderr
        raw = (tmp_path / "a.bin").read_bytes()
        w, h, depth, ctype, stride = struct.unpack("<5i", raw[:20])
        return (w, h, depth, ctype, np.frombuffer(raw[20:], np.uint8).reshape(h, stride)), ""

    n = 0
    for tag, data, px in _png_sweep():
        got, err = run(data)
        assert got is not None, (tag, err)
        w, h, depth, ctype, rows = got
        ref = PIL.open(io.BytesIO(data))
        if depth == 8:
            assert np.array_equal(rows, np.asarray(ref).reshape(h, -1)), tag               # palette images: the indices
        else:
            per = 8 // depth
            x = np.arange(w)
            samples = (rows[:, x // per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1)
            want = np.asarray(ref).astype(np.uint8) if ref.mode in ("P", "1") else None
            if want is not None:
                assert np.array_equal(samples, want), tag
            else:                                                                               # 4-bit grey: Pillow hands back the expanded value
                assert np.array_equal(samples * 255 // ((1 << depth) - 1), np.asarray(ref.convert("L"))), tag
        n += 1
    assert n == 84
    rgb = bytes(png_cases["rgb8_png"])
    for bad, word in ((_png_with(rgb, interlace=1), "interlaced"), (_png_with(rgb, depth=16), "16-bit"), (_png_with(rgb, ctype=4), "alpha"),
                      (rgb[:len(rgb) // 2], "truncated"), (rgb[:60] + bytes([rgb[60] ^ 0x40]) + rgb[61:], "CRC")):
        got, err = run(bad)
        assert got is None and word in err, (word, err)


def test_oracle_refuses_what_it_does_not_decode(cases, oracle):
    with pytest.raises(ValueError):
        oracle.jpeg_decode(b"\x89PNG\r\n\x1a\n" + bytes(64))
    data = bytes(cases["c420_q75_jpeg"])
    with pytest.raises(ValueError):
        oracle.jpeg_decode(data[:len(data) // 8])                     # truncated before the scan
    PIL = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    PIL.fromarray(np.zeros((32, 32, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        oracle.jpeg_decode(b.getvalue())


def test_oracle_bayer_known_answers(oracle):
    # a flat mosaic decodes to its three levels
    h, w = 12, 16
    m = np.zeros((h, w), np.uint8)
    m[0::2, 0::2] = 30; m[0::2, 1::2] = 120; m[1::2, 0::2] = 120; m[1::2, 1::2] = 200          # B G / G R
    out = oracle.bayer_bggr2bgr(m)
    assert np.all(out[..., 0] == 30) and np.all(out[..., 1] == 120) and np.all(out[..., 2] == 200)
    # interior formulas on a random mosaic, borders copy the neighbour
    rng = np.random.default_rng(5)
    m = rng.integers(0, 256, (9, 11)).astype(np.uint8)
    o = oracle.bayer_bggr2bgr(m).astype(int)
    mi = m.astype(int)
    y, x = 2, 2                                                        # a blue site
    assert o[y, x, 0] == mi[y, x] and o[y, x, 1] == (mi[y-1, x] + mi[y+1, x] + mi[y, x-1] + mi[y, x+1] + 2) >> 2
    assert o[y, x, 2] == (mi[y-1, x-1] + mi[y-1, x+1] + mi[y+1, x-1] + mi[y+1, x+1] + 2) >> 2
    y, x = 3, 4                                                        # green on a red row: red left/right, blue above/below
    assert o[y, x, 1] == mi[y, x] and o[y, x, 2] == (mi[y, x-1] + mi[y, x+1] + 1) >> 1 and o[y, x, 0] == (mi[y-1, x] + mi[y+1, x] + 1) >> 1
    assert np.array_equal(o[0], o[1]) and np.array_equal(o[-1], o[-2]) and np.array_equal(o[:, 0], o[:, 1]) and np.array_equal(o[:, -1], o[:, -2])


# ------------------------------------------------------------------ HIP path
@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 4096)
    yield c
    c.close()


@pytest.mark.gpu
def test_hip_jpeg_equals_libjpeg_turbo_fixture(cases, ctx):
    for name in cases["names"]:
        got = ctx.decode_image(bytes(cases[f"{name}_jpeg"]))
        assert np.array_equal(got, _bgr(cases[f"{name}_rgb"])), name


@pytest.mark.gpu
def test_hip_jpeg_full_hd_and_feeds_get_image(ctx, oracle):
    """A 1080p camera frame through the whole ingest: decode on the device, the BGR result handed to get_image without leaving HBM."""
    PIL = pytest.importorskip("PIL.Image")
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    scene = synth.Scene(7, 1920)
    g = synth.mono_frame(scene, 0, 1920, 1080)
    rgb = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0)], -1)
    for ss in (2, 1, 0):
        b = io.BytesIO()
        PIL.fromarray(rgb).save(b, "JPEG", quality=88, subsampling=ss)
        want = oracle.jpeg_decode(b.getvalue())
        assert np.array_equal(want, _bgr(np.asarray(PIL.open(io.BytesIO(b.getvalue())))))          # oracle == libjpeg-turbo at full size
        got = ctx.decode_image(b.getvalue())
        assert np.array_equal(got, want), ss
    dev = ctx.decode_image(b.getvalue(), device_out=True)
    K = synth.stereo_rig(1920).K_left
    Ks, newK, _ = uvo.resize_camera_matrix(1920, 1080, 640, K, np.zeros(4))
    a = ctx.get_image(dev, 640, Ks, np.zeros(4), newK, True, 3)
    assert np.array_equal(a, oracle.get_image(want, 640, Ks, np.zeros(4), newK, True, 3))
    with pytest.raises(uvo.UvoError):
        ctx.decode_image(b"not a jpeg at all")


@pytest.mark.gpu
def test_hip_png_equals_pillow_fixture_and_live(png_cases, ctx, oracle):
    """PNG payloads through uvo_decode_image: host inflate + filters, k_png_expand on the device; the size query parses headers only."""
    import ergo_uvo_amd as uvo
    for name in png_cases["names"]:
        got = ctx.decode_image(bytes(png_cases[f"{name}_png"]), "rgb8; png compressed")
        assert got.shape == png_cases[f"{name}_px"].shape and np.array_equal(got, png_cases[f"{name}_px"]), name
    for tag, data, px in _png_sweep():
        got = ctx.decode_image(data, "png")
        assert got.shape == px.shape and np.array_equal(got, px), tag
    # full HD, into device memory, and a bayer mosaic carried as a grey PNG (MU:161-164)
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(21)
    rgb = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    rgb[200:800] = (rgb[200:800] // 64) * 64
    b = io.BytesIO(); PIL.fromarray(rgb).save(b, "PNG", compress_level=3)
    dev = ctx.decode_image(b.getvalue(), "rgb8; png compressed", device_out=True)
    assert np.array_equal(dev.cpu().numpy(), rgb[..., ::-1])
    mosaic = rng.integers(0, 256, (96, 130), dtype=np.uint8)
    b = io.BytesIO(); PIL.fromarray(mosaic).save(b, "PNG")
    assert np.array_equal(ctx.decode_image(b.getvalue(), "bayer_bggr8; png compressed bayer_bggr8"), oracle.bayer_bggr2bgr(mosaic))
    good = bytes(png_cases["rgb8_png"])
    for bad in (_png_with(good, interlace=1), _png_with(good, depth=16), _png_with(good, ctype=4), good[:len(good) // 2],
                good[:60] + bytes([good[60] ^ 0x40]) + good[61:]):
        with pytest.raises(uvo.UvoError):
            ctx.decode_image(bad, "rgb8; png compressed")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 3), (9, 11), (240, 321), (1080, 1920)])
def test_hip_bayer_equals_oracle(ctx, oracle, shape):
    m = np.random.default_rng(11).integers(0, 256, shape).astype(np.uint8)
    assert np.array_equal(ctx.bayer_bggr2bgr(m), oracle.bayer_bggr2bgr(m))


@pytest.mark.gpu
def test_hip_bayer_jpeg_message(ctx, oracle):
    """A "bayer_bggr8; jpeg compressed" message: the grey JPEG is the mosaic, demosaiced after decoding (MU:161-164)."""
    PIL = pytest.importorskip("PIL.Image")
    m = np.random.default_rng(12).integers(0, 256, (120, 160)).astype(np.uint8)
    b = io.BytesIO()
    PIL.fromarray(m).save(b, "JPEG", quality=95)
    mosaic = oracle.jpeg_decode(b.getvalue())
    got = ctx.decode_image(b.getvalue(), "bayer_bggr8; jpeg compressed bayer_bggr8")
    assert got.shape == (120, 160, 3) and np.array_equal(got, oracle.bayer_bggr2bgr(mosaic))


@pytest.mark.gpu
def test_damaged_streams_are_refused_or_decoded_never_fatal(cases):
    """A camera topic can carry anything: 60 mutations per fixture (byte flips, truncations, injected markers, inserted bytes) go
    through uvo_decode_image; each returns an image or UvoError, and the context still decodes a good stream afterwards.  (The
    same mutations run through the oracle's decoder under ASan + UBSan in tests/test_oracle_sanitizers.py.)"""
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 640, 480, 256)
    frng = np.random.default_rng(99)
    n_ok = n_bad = 0
    try:
        for name in cases["names"]:
            base = bytearray(bytes(cases[f"{name}_jpeg"]))
            for trial in range(60):
                d = bytearray(base)
                kind = trial % 4
                if kind == 0:
                    for _ in range(1 + trial // 8):
                        d[int(frng.integers(2, len(d)))] = int(frng.integers(0, 256))
                elif kind == 1:
                    d = d[: int(frng.integers(2, len(d)))]
                elif kind == 2:
                    pos = int(frng.integers(2, len(d) - 4)); d[pos:pos + 2] = bytes([0xFF, int(frng.integers(0xC0, 0xFF))])
                else:
                    pos = int(frng.integers(2, len(d))); d[pos:pos] = bytes(frng.integers(0, 256, int(frng.integers(1, 40)), dtype=np.uint8))
                try:
                    img = c.decode_image(bytes(d), "jpeg"); n_ok += 1
                    assert img.size <= (1 << 26) * 3
                except uvo.UvoError:
                    n_bad += 1
            good = c.decode_image(bytes(base), "jpeg")
            assert np.array_equal(good, _bgr(cases[f"{name}_rgb"])), name
        assert n_bad > 50 and n_ok + n_bad == 60 * len(cases["names"])
    finally:
        c.close()
