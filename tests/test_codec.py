"""Compressed-image ingest: from_ros_to_cv_image = cv::imdecode (libjpeg) + COLOR_BayerBGGR2BGR
(uvo_libraries/src/math_utility.cpp:154-173).

The JPEG decode is the one component whose oracle is PINNED to a real third-party implementation: tests/golden/jpeg_cases.npz
holds JPEG streams and the pixels libjpeg-turbo (Pillow's) decoded from them; the oracle and the HIP path must reproduce them
byte for byte.  When Pillow is importable the CPU test also cross-checks a sweep of sizes / samplings / qualities live.
Bayer demosaicing has no third-party witness here (unpinned, recalled OpenCV behaviour): HIP <-> oracle + analytic cases."""
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
    with pytest.raises(uvo.UvoError):
        ctx.decode_image(b.getvalue(), "rgb8; png compressed")


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
