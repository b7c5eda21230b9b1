"""detect_features' ORB branch (VO_utility.cpp:100-105: ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10)->detectAndCompute) on the
GPU, through the C ABI (uvo_orb_detect), against the CPU oracle (oracle/o_orb.c) -- PARITY vs OpenCV UNPINNED; the oracle's parts are held
to independent numpy statements by tests/test_oracle_orb_kat.py.  Bit-exact: every pyramid level, its FAST score map and its blurred copy,
the keypoints (position, size, angle, Harris response, level) in OpenCV's order -- level by level, within a level the order
KeyPointsFilter::retainBest leaves --, the 32-byte rBRIEF rows; then the rows through the Hamming matcher of match_features
(VO_utility.cpp:520-524), the ctypes mirror's detect_features("ORB") and the C++ surface (detect_features with FEATURE_DETECTOR = "ORB").

The sampling table is an INPUT of the detector (OpenCV's learned bit_pattern_31_ cannot be restated): the tests pass the table OpenCV's
own makeRandomPattern draws (cv::RNG(0x34985739), restated in the oracle), which exercises the same code."""
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


@pytest.mark.parametrize("w,h,seed", [(640, 360, 77), (641, 363, 78), (320, 200, 79), (1280, 720, 80), (1920, 1080, 81), (150, 120, 82)])
def test_orb_detect_and_compute_bit_exact(oracle, w, h, seed):
    """The reference's arguments at six sizes: odd sizes (the resize tables' edge entries), 1080p (every level's share is cut twice by
    retainBest), and 150 x 120 (the top levels are smaller than twice the edge margin: no keypoints there)."""
    import ergo_uvo_amd as uvo
    img = _scene(w, h, seed)
    pat = oracle.orb_random_pattern()
    ctx = uvo.Context(uvo.Params.stereo(), 0, w, h, 4096)
    try:
        ctx.orb_set_pattern(pat)
        kps, desc = ctx.orb_detect(img, cap=1 << 16)
        ko, do = oracle.orb_detect(img, pat, cap=1 << 17)
        for level in (1, 4, 7):
            got = ctx.orb_plane(level, 0)
            assert np.array_equal(got, oracle.orb_level_image(img, level)), level
            assert np.array_equal(ctx.orb_plane(level, 1), oracle.orb_level_image(img, level, blurred=True)), level
            assert np.array_equal(ctx.orb_plane(level, 2), oracle.fast_scores(got, 10)), level
        assert np.array_equal(ctx.orb_plane(0, 2), oracle.fast_scores(img, 10))
        assert len(ko) > (20 if w < 200 else 1000)
        _same_kps(kps, ko)
        assert desc.shape == (len(ko), 32) and np.array_equal(desc, do)
        k2, d2 = ctx.orb_detect(img, cap=1 << 16, descriptors=False)                # keypoints only
        assert d2 is None
        _same_kps(k2, ko)
    finally:
        ctx.close()


def test_orb_other_arguments_and_device_images(oracle):
    """uvo_orb_configure: fewer features (both rankings cut hard, many ties among the integer FAST scores), another scale factor and
    FAST threshold, five levels; a device-resident image with a row pitch."""
    import torch
    import ergo_uvo_amd as uvo
    w, h = 800, 450
    img = _scene(w, h, 83)
    pat = oracle.orb_random_pattern()
    ctx = uvo.Context(uvo.Params.stereo(), 0, w, h, 4096)
    try:
        ctx.orb_set_pattern(pat)
        for kw in (dict(nfeatures=500), dict(nfeatures=1500, scaleFactor=1.3, nlevels=5, fastThreshold=25), dict(nfeatures=40, nlevels=2, edgeThreshold=40)):
            ctx.orb_configure(**kw)
            kps, desc = ctx.orb_detect(img, cap=1 << 16)
            ko, do = oracle.orb_detect(img, pat, **kw)
            assert len(ko) >= kw["nfeatures"] * 0.9
            _same_kps(kps, ko)
            assert np.array_equal(desc, do)
        ctx.orb_configure()
        dev = torch.from_numpy(img).cuda()
        kps, desc = ctx.orb_detect(dev, cap=1 << 16)
        ko, do = oracle.orb_detect(img, pat)
        _same_kps(kps, ko)
        assert np.array_equal(desc, do)
    finally:
        ctx.close()


def test_orb_rows_through_the_hamming_matcher_and_the_mirror(oracle):
    """match_features' ORB arm (VO_utility.cpp:520-524: BFMatcher(NORM_HAMMING) + the ratio test) on the rows of two views of one scene."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    L, R = synth.stereo_pair(synth.Scene(91, 640), 0, 640, 360)
    pat = oracle.orb_random_pattern()
    ctx = uvo.Context(uvo.Params.stereo(), 0, 640, 360, 16384)
    try:
        ctx.set_feature_detector("ORB")
        with pytest.raises(uvo.UvoError, match="bit_pattern_31_"):
            ctx.detect_features(L)                                             # no table yet: said, not guessed
        ctx.orb_set_pattern(pat)
        k1, d1 = ctx.detect_features(L)                                        # the mirror's detect_features, FEATURE_DETECTOR == "ORB"
        k2, d2 = ctx.detect_features(R)
        assert d1.dtype == np.uint8 and d1.shape[1] == 32 and len(k1) > 2000 and len(k2) > 2000
        m = ctx.match_features_hamming(d1, d2, ratio=0.8)
        o1, od1 = oracle.orb_detect(L, pat); o2, od2 = oracle.orb_detect(R, pat)
        mo = oracle.match_hamming(od1, od2, 0.8)
        assert np.array_equal(m["queryIdx"], mo["queryIdx"]) and np.array_equal(m["trainIdx"], mo["trainIdx"]) and np.array_equal(m["distance"], mo["distance"])
        assert len(m) > 300
        dy = k1["y"][m["queryIdx"]] - k2["y"][m["trainIdx"]]
        dx = k1["x"][m["queryIdx"]] - k2["x"][m["trainIdx"]]
        assert np.mean((np.abs(dy - np.median(dy)) < 3.0) & (dx > 0)) > 0.85    # true matches share one vertical offset and have a positive disparity
        with pytest.raises(uvo.UvoError):
            ctx._check(ctx._lib.uvo_ctx_set_feature_detector(ctx._h, b"ORB"))     # the fused steps run on SURF or SIFT: said, not silently ignored
    finally:
        ctx.close()


def test_shim_detect_features_orb_branch(tmp_path, oracle):
    """detect_features(img, keypoints, descriptors) with FEATURE_DETECTOR = "ORB" through the C++ surface: CV_8U rows of 32 bytes; the table from
    UVO_ORB_PATTERN_FILE (a text file of the 1024 integers, as they stand in OpenCV's source); without it the branch throws."""
    from ergo_uvo_amd import KP_DTYPE
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ergo_uvo_amd", "shim"), "-s"])
    img = _scene(640, 360, 77)
    pat = oracle.orb_random_pattern()
    inp, outp, patf = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "bit_pattern_31.txt"
    inp.write_bytes(struct.pack("<iii8s", 640, 360, 1500, b"ORB") + img.tobytes())
    patf.write_text(",\n".join(", ".join(str(int(v)) for v in row) for row in pat.reshape(256, 4)) + "\n")
    exe = os.path.join(ROOT, "tests", "cpp", "build", "shim_detect")
    env = dict(os.environ); env.pop("UVO_ORB_PATTERN_FILE", None)
    res = subprocess.run([exe, str(inp), str(outp)], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 4 and "bit_pattern_31_" in res.stderr
    env["UVO_ORB_PATTERN_FILE"] = str(patf)
    res = subprocess.run([exe, str(inp), str(outp)], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr
    raw = outp.read_bytes()
    n, cols = struct.unpack("<ii", raw[:8])
    kps = np.frombuffer(raw[8:8 + 28 * n], KP_DTYPE)
    desc = np.frombuffer(raw[8 + 28 * n:], np.uint8).reshape(n, cols)
    ko, do = oracle.orb_detect(img, pat)
    assert cols == 32
    _same_kps(kps, ko)
    assert np.array_equal(desc, do)


def test_orb_misuse(oracle):
    import ergo_uvo_amd as uvo
    ctx = uvo.Context(uvo.Params.stereo(), 0, 320, 200, 64)
    try:
        img = _scene(320, 200, 79)
        with pytest.raises(uvo.UvoError, match="bit_pattern_31_"):
            ctx.orb_detect(img)                                                # descriptors without a table
        k, d = ctx.orb_detect(img, cap=1 << 15, descriptors=False)
        assert len(k) > 500 and d is None
        with pytest.raises(uvo.UvoError):
            ctx.orb_detect(img, cap=16, descriptors=False)                     # output capacity: an error, not a truncated list
        with pytest.raises(uvo.UvoError):
            ctx.orb_detect(np.zeros((400, 400), np.uint8), descriptors=False)  # larger than the context
        bad = oracle.orb_random_pattern().copy(); bad[7, 0] = 16
        with pytest.raises(uvo.UvoError):
            ctx.orb_set_pattern(bad)                                           # a coordinate outside the patch
        with pytest.raises(uvo.UvoError):
            ctx.orb_configure(edgeThreshold=10)                                # the rotated table would reach outside the image
        k, d = ctx.orb_detect(np.full((200, 320), 90, np.uint8), cap=64, descriptors=False)   # a blank image: no corners
        assert len(k) == 0
    finally:
        ctx.close()
