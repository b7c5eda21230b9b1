"""detect_features' AKAZE branch (VO_utility.cpp:93-98: AKAZE::create()->detectAndCompute) on the GPU, through the C ABI
(uvo_akaze_detect), against the CPU oracle (oracle/o_akaze.c) -- PARITY vs OpenCV UNPINNED; the oracle's parts are held to their closed
forms by tests/test_oracle_akaze_kat.py.  Bit-exact: every plane of the non-linear scale space and of the Hessian response, the
keypoints (position, size, angle, response, level) in OpenCV's order, the 61-byte M-LDB rows; then the rows through the Hamming
matcher of match_features (VO_utility.cpp:520-524), the ctypes mirror's detect_features("AKAZE"), and the C++ surface
(detect_features with FEATURE_DETECTOR = "AKAZE")."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(w, h, seed):
    from ergo_uvo_amd import synth
    return synth.stereo_pair(synth.Scene(seed, w), 0, w, h)[0]


def _same_kps(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in a.dtype.names:
        av, bv = a[f], b[f]
        assert np.array_equal(av.view(np.uint32) if av.dtype.kind == "f" else av, bv.view(np.uint32) if bv.dtype.kind == "f" else bv), f


@pytest.mark.parametrize("w,h,seed", [(640, 360, 77), (641, 363, 78), (320, 200, 79), (1280, 720, 80), (1920, 1080, 81), (120, 90, 82)])
def test_akaze_detect_and_compute_bit_exact(oracle, w, h, seed):
    """Four octaves at even sizes (exact halving), odd sizes (641 x 363: the general INTER_AREA tables), one octave only (120 x 90), 720p and 1080p."""
    import ergo_uvo_amd as uvo
    img = _scene(w, h, seed)
    ctx = uvo.Context(uvo.Params.stereo(), 0, w, h, 32768)
    try:
        kps, desc = ctx.akaze_detect(img)
        ko, do = oracle.akaze_detect(img, cap=1 << 17)
        lv, _ = oracle.akaze_levels(w, h)
        for level in sorted({0, 1, len(lv) // 2, len(lv) - 1}):
            for what in (0, 4) if w > 700 else (0, 1, 2, 3, 4):
                got = ctx.akaze_plane(level, what)
                want, _ = oracle.akaze_plane(img, level, what)
                assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (level, what, np.abs(got - want).max())
        assert len(ko) > (5 if w < 200 else 150)
        _same_kps(kps, ko)
        assert desc.shape == (len(ko), 61) and np.array_equal(desc, do)
    finally:
        ctx.close()


def test_akaze_rows_through_the_hamming_matcher_and_the_mirror(oracle):
    """match_features' AKAZE arm (VO_utility.cpp:520-524: BFMatcher(NORM_HAMMING) + the ratio test) on the rows of two views of one scene:
    the oracle's matcher on the oracle's rows gives the same list; most matches join keypoints that the known disparity relates."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    L, R = synth.stereo_pair(synth.Scene(91, 640), 0, 640, 360)
    ctx = uvo.Context(uvo.Params.stereo(), 0, 640, 360, 16384)
    try:
        ctx.set_feature_detector("AKAZE")
        k1, d1 = ctx.detect_features(L)                                        # the mirror's detect_features, FEATURE_DETECTOR == "AKAZE"
        k2, d2 = ctx.detect_features(R)
        assert d1.dtype == np.uint8 and d1.shape[1] == 61 and len(k1) > 300 and len(k2) > 300
        m = ctx.match_features_hamming(d1, d2, ratio=0.8)
        o1, od1 = oracle.akaze_detect(L); o2, od2 = oracle.akaze_detect(R)
        mo = oracle.match_hamming(od1, od2, 0.8)
        assert np.array_equal(m["queryIdx"], mo["queryIdx"]) and np.array_equal(m["trainIdx"], mo["trainIdx"]) and np.array_equal(m["distance"], mo["distance"])
        assert len(m) > 100
        dy = k1["y"][m["queryIdx"]] - k2["y"][m["trainIdx"]]
        dx = k1["x"][m["queryIdx"]] - k2["x"][m["trainIdx"]]
        # the rig's two cameras differ by a horizontal baseline (and their principal points): true matches share one vertical offset
        # and have a positive disparity
        assert np.mean((np.abs(dy - np.median(dy)) < 2.0) & (dx > 0)) > 0.9
        with pytest.raises(uvo.UvoError):
            ctx._check(ctx._lib.uvo_ctx_set_feature_detector(ctx._h, b"AKAZE"))   # the fused steps run on SURF or SIFT: said, not silently ignored
    finally:
        ctx.close()


def test_shim_detect_features_akaze_branch(tmp_path, oracle):
    """detect_features(img, keypoints, descriptors) with FEATURE_DETECTOR = "AKAZE" through the C++ surface: CV_8U rows of 61 bytes."""
    from ergo_uvo_amd import KP_DTYPE
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ergo_uvo_amd", "shim"), "-s"])
    img = _scene(640, 360, 77)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(struct.pack("<iii8s", 640, 360, 1500, b"AKAZE") + img.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_detect"), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    raw = outp.read_bytes()
    n, cols = struct.unpack("<ii", raw[:8])
    kps = np.frombuffer(raw[8:8 + 28 * n], KP_DTYPE)
    desc = np.frombuffer(raw[8 + 28 * n:], np.uint8).reshape(n, cols)
    ko, do = oracle.akaze_detect(img)
    assert cols == 61
    _same_kps(kps, ko)
    assert np.array_equal(desc, do)


def test_akaze_misuse(oracle):
    import ergo_uvo_amd as uvo
    ctx = uvo.Context(uvo.Params.stereo(), 0, 320, 200, 64)
    try:
        img = _scene(320, 200, 79)
        with pytest.raises(uvo.UvoError):
            ctx.akaze_detect(img)                                              # more keypoints than max_kpts: a capacity error, not a truncated list
        with pytest.raises(uvo.UvoError):
            ctx.akaze_detect(np.zeros((400, 400), np.uint8))                   # larger than the context
        k, d = ctx.akaze_detect(np.full((200, 320), 90, np.uint8))             # a blank image: no keypoints, kcontrast falls back to 0.03
        assert len(k) == 0 and d.shape == (0, 61)
    finally:
        ctx.close()
