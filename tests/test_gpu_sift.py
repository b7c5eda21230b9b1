"""detect_features' "SIFT" branch (VO_utility.cpp:107-112: SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute) on the GPU against
the CPU oracle (oracle/o_sift.c), through the C ABI (uvo_sift_detect) and through the C++ uvo_libraries surface.

Bit-exact: keypoints (all seven fields) and the 128 descriptor entries, every pyramid layer.  The oracle is the builder's
restatement of OpenCV 4.5's scalar paths; parity against OpenCV itself is unpinned (o_sift.c header)."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 1920, 1080, 8192)
    yield c
    c.close()


def scene_image(w, h, seed=123, k=0):
    from ergo_uvo_amd import synth
    return synth.stereo_pair(synth.Scene(seed, w), k, w, h)[0]


def same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def check(ctx, oracle, img, **kw):
    kg, dg = ctx.sift_detect(img, **kw)
    ko, do = oracle.sift_detect(np.asarray(img.cpu() if hasattr(img, "cpu") else img), **kw)
    assert len(kg) == len(ko), (len(kg), len(ko))
    assert same(kg, ko), np.where(kg.view(np.uint8).reshape(len(kg), -1) != ko.view(np.uint8).reshape(len(ko), -1))[0][:5]
    assert same(dg, do), np.where((dg != do).any(axis=1))[0][:5]
    return kg, dg


@pytest.mark.parametrize("w,h", [(203, 131), (97, 64), (640, 360)])
def test_pyramid_layers_bit_exact(ctx, oracle, w, h):
    """createInitialImage + buildGaussianPyramid + buildDoGPyramid: every Gaussian layer of every octave, and the differences."""
    img = scene_image(640, 360)[:h, :w].copy()
    ctx.sift_detect(img)
    n_oct = int(np.rint(np.log2(2 * min(w, h)) - 2)) + 1
    for o in range(n_oct):
        prev = None
        for l in range(6):
            g = ctx.sift_layer(o, l)
            og = oracle.sift_gauss_layer(img, o, l)
            assert same(g, og), (o, l, float(np.max(np.abs(g - og))))
            if prev is not None:
                assert same(ctx.sift_layer(o, l - 1, dog=True), og - prev), (o, l)
            prev = og
    with pytest.raises(Exception):
        ctx.sift_layer(n_oct, 0)


@pytest.mark.parametrize("w,h,seed", [(640, 360, 123), (203, 131, 7), (97, 64, 9), (1280, 720, 11)])
def test_sift_detect_bit_exact(ctx, oracle, w, h, seed):
    img = scene_image(max(w, 320), max(h, 180), seed)[:h, :w].copy()
    kg, dg = check(ctx, oracle, img)
    assert len(kg) > 20
    octave = (kg["octave"] & 255).astype(np.int64); octave[octave >= 128] -= 256
    assert octave.min() == -1 and octave.max() >= 2       # several octaves took part


def test_sift_detect_1080p_and_retain_best(ctx, oracle):
    """The headline geometry: more than 10000 keypoints survive at 1920 x 1080, so retainBest(10000) is active (VOU:109)."""
    img = scene_image(1920, 1080, 20250910)
    kg, dg = check(ctx, oracle, img)
    assert len(kg) >= 10000
    k_all, _ = ctx.sift_detect(img, nfeatures=0, cap=1 << 16)
    assert len(k_all) > len(kg)
    cut = np.sort(k_all["response"])[::-1][9999]
    assert same(kg, k_all[k_all["response"] >= cut])


@pytest.mark.parametrize("kw", [dict(nfeatures=300), dict(n_octave_layers=2), dict(n_octave_layers=5, sigma=1.2), dict(contrast_threshold=0.08, edge_threshold=4.0),
                                dict(nfeatures=0, contrast_threshold=0.01)])
def test_sift_parameters(ctx, oracle, kw):
    img = scene_image(640, 360, 31)
    check(ctx, oracle, img, **kw)


def test_sift_blobs_flat_and_noise(ctx, oracle):
    y, x = np.mgrid[0:120, 0:160]
    for s in (3.0, 8.0):
        img = np.clip(40 + 160 * np.exp(-((x - 70.3) ** 2 + (y - 55.6) ** 2) / (2 * s * s)), 0, 255).round().astype(np.uint8)
        kg, _ = check(ctx, oracle, img)
        assert len(kg) >= 1 and abs(kg["x"][0] - 70.55) < 0.08
    flat = np.full((64, 80), 77, np.uint8)
    kg, dg = check(ctx, oracle, flat)
    assert len(kg) == 0 and dg.shape == (0, 128)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (150, 210), dtype=np.uint8)      # dense extrema, many refinements that wander and are dropped
    check(ctx, oracle, noise, nfeatures=0)
    check(ctx, oracle, noise)


def test_sift_from_device_and_strided_images(ctx, oracle):
    import torch
    img = scene_image(640, 360, 5)
    k0, d0 = check(ctx, oracle, img)
    kd, dd = ctx.sift_detect(torch.from_numpy(img).cuda())
    assert same(kd, k0) and same(dd, d0)
    # a padded host image through the C ABI directly (row pitch > width), and a window of a larger device image
    import ctypes as C
    from ergo_uvo_amd import KP_DTYPE
    pad = np.zeros((360, 704), np.uint8); pad[:, :640] = img
    n = C.c_int(0)
    kps = np.zeros(16384, KP_DTYPE); desc = np.zeros((16384, 128), np.float32)
    ctx._check(ctx._lib.uvo_sift_detect(ctx._h, pad.ctypes.data_as(C.c_void_p), 640, 360, 704, 0, 10000, 3, 0.03, 10.0, 1.6,
                                        kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 16384, C.byref(n)))
    assert same(kps[:n.value], k0) and same(desc[:n.value], d0)
    # keypoints only / count only
    ctx._check(ctx._lib.uvo_sift_detect(ctx._h, pad.ctypes.data_as(C.c_void_p), 640, 360, 704, 0, 10000, 3, 0.03, 10.0, 1.6, None, None, 0, C.byref(n)))
    assert n.value == len(k0)
    # the SURF path of the same context is untouched by the SIFT workspace
    ks, ds = ctx.detect_features(img)
    ko, do = oracle.surf(img, 1500.0)
    assert same(ks, ko) and same(ds, do)


def test_sift_refusals(ctx):
    import ctypes as C
    img = scene_image(640, 360, 5)
    with pytest.raises(Exception, match="capacity"):
        ctx.sift_detect(img, cap=50)
    with pytest.raises(Exception):
        ctx.sift_detect(img, n_octave_layers=0)
    with pytest.raises(Exception):
        ctx.sift_detect(img, sigma=0.4)
    with pytest.raises(Exception, match="63 taps"):
        ctx.sift_detect(img, n_octave_layers=1)           # its last blur has sigma 11: 91 taps
    with pytest.raises(Exception):
        ctx.sift_detect(np.zeros((8, 8), np.uint8))
    k, d = ctx.sift_detect(img)                           # still usable afterwards
    assert len(k) > 100


def _shim_detect(tmp_path, img, name, min_hessian=1500):
    from ergo_uvo_amd import KP_DTYPE
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ergo_uvo_amd", "shim"), "-s"])
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    h, w = img.shape
    inp.write_bytes(struct.pack("<iii8s", w, h, min_hessian, name.encode()) + img.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_detect"), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    if res.returncode != 0:
        return res.returncode, res.stderr, None
    raw = outp.read_bytes()
    n, cols = struct.unpack("<ii", raw[:8])
    kps = np.frombuffer(raw[8:8 + 28 * n], KP_DTYPE)
    body = raw[8 + 28 * n:]
    desc = np.frombuffer(body, np.uint8).reshape(n, cols) if len(body) == n * cols else np.frombuffer(body, np.float32).reshape(n, cols)   # AKAZE rows are bytes
    return 0, kps, desc


def test_shim_detect_features_sift_and_surf_branches(tmp_path, oracle):
    """detect_features(img, keypoints, descriptors) with FEATURE_DETECTOR = "SIFT" / "SURF" through the C++ surface; "ORB" is refused."""
    img = scene_image(640, 360, 17)
    rc, kps, desc = _shim_detect(tmp_path, img, "SIFT")
    assert rc == 0, kps
    ko, do = oracle.sift_detect(img)
    assert desc.shape[1] == 128 and same(kps, ko) and same(desc, do)
    rc, kps, desc = _shim_detect(tmp_path, img, "SURF")
    assert rc == 0 and desc.shape[1] == 64 and len(kps) > 100
    rc, err, _ = _shim_detect(tmp_path, img, "ORB")
    assert rc == 4 and "FEATURE_DETECTOR" in err                     # (the "AKAZE" branch: tests/test_gpu_akaze.py)


# ---------------------------------------------------------------------------------------------------------------------------------
# FEATURE_DETECTOR = "SIFT" inside the fused steps (uvo_ctx_set_feature_detector): uvo_stereo_step / submit / collect and uvo_mono_step
# take detect_features' SIFT branch and match 128-float rows; the oracle's state machines are switched to the same detector.
@pytest.fixture()
def sctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 8192)
    c.set_feature_detector("SIFT")
    yield c
    c.close()


def _stereo_fields(r):
    return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers)


def test_fused_stereo_step_on_sift_matches_oracle(sctx, oracle, scene_small):
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    seq = [scene_small[k] for k in (0, 1, 2, 1, 0)]
    sctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    ovo.use_sift()
    sync = []
    for k, (L, R) in enumerate(seq):
        r = sctx.stereo_step(L, R, 0.05)
        o = ovo.step(L, R, 0.05)
        assert _stereo_fields(r) == _stereo_fields(o), (k, _stereo_fields(r), _stereo_fields(o))
        for what in ("kps_left", "kps_right", "desc_left", "desc_right", "matches_stereo", "matches_tri", "good_idx", "inliers"):
            a, b = sctx.stereo_get(what), ovo.get(what)
            assert same(a, b), (k, what, a.shape, b.shape)
        for a, b in ((r.rvec, o.rvec), (r.tvec, o.tvec), (r.t_prev_curr, o.t_prev_curr)):
            a, b = np.array(list(a)), np.array(list(b))
            assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12), (k, a, b)
        sync.append((_stereo_fields(r), tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr)))
    assert sum(f[0][0] for f in sync) == len(seq) - 1 and sync[-1][0][2] > 1000 and sctx.stereo_get("desc_left").shape[1] == 128
    # the same sequence with several pairs in flight
    for depth in (3, 2):
        sctx.stereo_set_depth(depth)
        sctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        piped, sub = [], 0
        sctx.stereo_submit(*seq[0]); sub += 1
        r = sctx.stereo_collect(0.05); piped.append((_stereo_fields(r), tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr)))
        while len(piped) < len(seq):
            while sub < len(seq) and sub - len(piped) < depth:
                sctx.stereo_submit(*seq[sub]); sub += 1
            r = sctx.stereo_collect(0.05); piped.append((_stereo_fields(r), tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr)))
        assert piped == sync, depth


def test_fused_mono_step_on_sift_matches_oracle(oracle, mono_small):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    p = uvo.Params.mono(SURF_MIN_HESSIAN=400, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8)
    c = uvo.Context(p, 0, 640, 360, 8192)
    try:
        c.set_feature_detector("SIFT")
        c.mono_set_camera(rig.K_left)
        ovo = oracle.MonoVO(oracle.mono_params(400, 8), rig.K_left)
        ovo.use_sift()
        nvalid = 0
        for k, img in enumerate([mono_small[i] for i in (0, 1, 2, 1)]):
            r = c.mono_step(img, 4.0, 0.05)
            o = ovo.step(img, 4.0, 0.05)
            for f in ("valid", "initialized", "n_kps", "n_matches", "n_inliers"):
                assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
            assert same(c.mono_get("kps"), ovo.get("kps")) and same(c.mono_get("matches"), ovo.get("matches")) and same(c.mono_get("mask"), ovo.get("mask"))
            nvalid += r.valid
        assert nvalid >= 2 and r.n_kps > 1000
        # the same frames with three in flight (uvo_mono_submit / collect)
        seq = [mono_small[i] for i in (0, 1, 2, 1, 0, 2)]
        fields = ("published", "valid", "initialized", "used_essential", "success", "n_kps", "n_matches", "n_inliers", "n_good3d", "n_front")
        c.mono_reset()
        want = []
        for img in seq:
            r = c.mono_step(img, 4.0, 0.2)
            want.append((tuple(getattr(r, f) for f in fields), tuple(r.R), tuple(r.t), c.mono_get("mask").copy()))
        c.mono_reset()
        c.stereo_set_depth(3)
        got, sub = [], 0
        for i in range(len(seq)):
            while sub < len(seq) and sub - i < 3:
                c.mono_submit(seq[sub], 4.0); sub += 1
            r = c.mono_collect(0.2)
            got.append((tuple(getattr(r, f) for f in fields), tuple(r.R), tuple(r.t), c.mono_get("mask").copy()))
        for k, (a, b) in enumerate(zip(want, got)):
            assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3]), k
    finally:
        c.close()


def test_feature_detector_switch_refusals(scene_small):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 1024)       # room for 1024 keypoints: SIFT finds ~1500 here
    try:
        with pytest.raises(uvo.UvoError, match="SURF.*SIFT"):
            c.set_feature_detector("BRISK")                                             # not one of the reference's four names
        with pytest.raises(uvo.UvoError, match="SURF.*SIFT"):
            c._check(c._lib.uvo_ctx_set_feature_detector(c._h, b"ORB"))                 # AKAZE / ORB are operators: the FUSED steps run on SURF or SIFT
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        c.set_feature_detector("SIFT")
        with pytest.raises(uvo.UvoError, match="max_kpts"):
            c.stereo_step(*scene_small[0], 0.05)
        k, d = c.detect_features(scene_small[0][0])          # the mirror of detect_features follows the switch (and is not bound by max_kpts)
        assert d.shape[1] == 128 and len(k) > 1024
        c.stereo_reset()
        c.set_feature_detector("SURF")
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        assert c.stereo_step(*scene_small[0], 0.05).n_left > 100
        assert c.stereo_step(*scene_small[1], 0.05).valid == 1
        with pytest.raises(uvo.UvoError, match="reset"):
            c.set_feature_detector("SIFT")                                              # a sequence is running on SURF descriptors
    finally:
        c.close()
    # lists too small for the frame (64 keypoints of room: 1024 candidates, 256 oriented keypoints): the same error, nothing written past them
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 64)
    try:
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        c.set_feature_detector("SIFT")
        with pytest.raises(uvo.UvoError, match="max_kpts"):
            c.stereo_step(*scene_small[0], 0.05)
        k, d = c.detect_features(scene_small[0][0])          # the standalone operator grows its lists instead
        assert len(k) > 1024
    finally:
        c.close()


def test_standalone_surf_operators_keep_their_row_width_when_the_loops_run_on_sift(scene_small):
    """with FEATURE_DETECTOR = "SIFT" the fused steps use 128-float rows, but uvo_surf_detect / uvo_match_knn2 /
    uvo_match_knn2_ratio (no `dim`) are this context's SURF operators: n x 64 buffers in, the same answers as before the switch."""
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 4096)
    try:
        L, R = scene_small[0]
        k1, d1 = c.surf_detect(L)
        k2, d2 = c.surf_detect(R)
        assert d1.shape[1] == 64 and len(k1) > 100
        idx0, dist0 = c.knn_match(d1, d2)
        m0 = c.match_features(d1, d2)
        c.set_feature_detector("SIFT")
        k1s, d1s = c.surf_detect(L)
        assert np.array_equal(k1s, k1) and np.array_equal(d1s, d1)
        idx1, dist1 = c.knn_match(d1, d2)                       # would have staged 128 * n floats out of n x 64 buffers
        assert np.array_equal(idx1, idx0) and np.array_equal(dist1, dist0)
        assert np.array_equal(c.match_features(d1, d2), m0)
        ks, ds = c.detect_features(L)                            # the reference's detect_features follows the switch
        assert ds.shape[1] == 128
        with pytest.raises(ValueError, match="dim=128"):
            c.knn_match(ds, ds)
        i128, _ = c.knn_match(ds[:500], ds[:500], dim=128)
        assert np.array_equal(i128[:, 0], np.arange(500)) or (i128[:, 0] != np.arange(500)).sum() < 5      # a row's nearest row is itself (or an exact duplicate)
    finally:
        c.close()


def test_fused_stereo_step_on_sift_1280x720(oracle):
    """A larger geometry through the same path (several thousand keypoints per image, every octave kind of the pyramid in use)."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    W, H = 1280, 720
    scene = synth.Scene(11, W)
    rig = synth.stereo_rig(W)
    seq = [synth.stereo_pair(scene, k, W, H) for k in (0, 1, 2)]
    c = uvo.Context(uvo.Params.stereo(), 0, W, H, 12288)
    try:
        c.set_feature_detector("SIFT")
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ovo.use_sift()
        for k, (L, R) in enumerate(seq):
            r = c.stereo_step(L, R, 0.05)
            o = ovo.step(L, R, 0.05)
            assert _stereo_fields(r) == _stereo_fields(o), (k, _stereo_fields(r), _stereo_fields(o))
            assert same(c.stereo_get("inliers"), ovo.get("inliers")) and same(c.stereo_get("matches_tri"), ovo.get("matches_tri"))
            a, b = np.array(list(r.tvec)), np.array(list(o.tvec))
            assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12)
        assert r.valid == 1 and r.n_left > 3000
    finally:
        c.close()


def test_fused_sift_pipeline_soak_is_periodic(sctx, scene_small):
    """300 pairs through four lanes on SIFT features, frames in a period-4 order: a pair's result depends on its frames and on the
    previous pair's "after stereo match" set only, so from the second period on every result must repeat the one a period earlier,
    bit for bit (a race between lanes, a stale workspace or a list overflow would break the repetition)."""
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    order = [0, 1, 2, 1]
    n = 300
    sctx.stereo_set_depth(4)
    sctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    res, sub = [], 0
    sctx.stereo_submit(*scene_small[order[0]]); sub += 1
    r = sctx.stereo_collect(0.05); res.append((_stereo_fields(r), tuple(r.rvec), tuple(r.tvec)))
    while len(res) < n:
        while sub < n and sub - len(res) < 4:
            sctx.stereo_submit(*scene_small[order[sub % 4]]); sub += 1
        r = sctx.stereo_collect(0.05); res.append((_stereo_fields(r), tuple(r.rvec), tuple(r.tvec)))
    assert all(f[0][0] == 1 for f in res[1:])
    for i in range(8, n):
        assert res[i] == res[i - 4], i


def test_fused_stereo_step_on_sift_odd_geometry(oracle, scene_small):
    """Odd width and height (no octave has a width that is a multiple of four, the stereo rig's principal point off the crop's
    centre): the same loop on a 417 x 243 window of the frames."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    seq = [(np.ascontiguousarray(L[50:293, 100:517]), np.ascontiguousarray(R[50:293, 100:517])) for L, R in (scene_small[0], scene_small[1], scene_small[2])]
    c = uvo.Context(uvo.Params.stereo(), 0, 417, 243, 8192)
    try:
        c.set_feature_detector("SIFT")
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ovo.use_sift()
        for k, (L, R) in enumerate(seq):
            r = c.stereo_step(L, R, 0.05)
            o = ovo.step(L, R, 0.05)
            assert _stereo_fields(r) == _stereo_fields(o), (k, _stereo_fields(r), _stereo_fields(o))
            for what in ("kps_left", "desc_right", "matches_stereo", "matches_tri", "inliers"):
                assert same(c.stereo_get(what), ovo.get(what)), (k, what)
        assert r.n_left > 300 and r.n_stereo_matches > 100
    finally:
        c.close()


def test_failure_paths_through_the_pipeline_on_sift(sctx, oracle, scene_small):
    """The gate ladder (VO:556/567/626/634/665) with SIFT features: frames without features in the middle of a sequence, the state kept
    on failure and the empty "after stereo match" sets handed on -- synchronously, with several pairs in flight, and in the oracle
    (the SURF twin of this test is test_gpu_parity.py::test_failure_paths_through_the_pipeline)."""
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    blank = (np.full_like(scene_small[0][0], 90), np.full_like(scene_small[0][1], 90))
    half = (scene_small[1][0], blank[1])                      # features on the left only: VO:556 fails on the right count
    seq = [scene_small[0], scene_small[1], blank, scene_small[2], scene_small[1], half, scene_small[0], scene_small[1], scene_small[2]]

    def fields(r):
        return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers,
                tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr), tuple(r.velocity))

    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    ovo.use_sift()
    want = [fields(ovo.step(L, R, 0.05)) for L, R in seq]
    assert [w[0] for w in want] == [0, 1, 0, 0, 1, 0, 0, 1, 1]       # init, ok, blank, no prev set, ok, half-blank, no prev set, ok, ok
    sctx.stereo_set_depth(1)
    sctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    sync = [fields(sctx.stereo_step(L, R, 0.05)) for L, R in seq]
    for g, e in zip(sync, want):
        assert g[:8] == e[:8], (g[:8], e[:8])
        for a, b in zip(g[8:], e[8:]):
            a, b = np.array(a), np.array(b)
            assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12), (a, b)
    for depth in (2, 4):
        sctx.stereo_set_depth(depth)
        sctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        piped, sub = [], 0
        sctx.stereo_submit(*seq[0]); sub += 1
        piped.append(fields(sctx.stereo_collect(0.05)))
        while len(piped) < len(seq):
            while sub < len(seq) and sub - len(piped) < depth:
                sctx.stereo_submit(*seq[sub]); sub += 1
            piped.append(fields(sctx.stereo_collect(0.05)))
        assert piped == sync, depth
