"""Parity of the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Bit-exact for integers, keypoints, descriptors, match pairs, masks; poses to 1e-4 rel. Frobenius
(BASELINE.json north_star), in practice bit-equal.  PARITY vs OpenCV itself is UNPINNED (no OpenCV
in this environment): what is shown is HIP <-> the oracle's restatement."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 20000)
    yield c
    c.close()


def _rand_img(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2)).astype(np.float64)
    from scipy import ndimage
    img = ndimage.zoom(base, 8, order=1)[:h, :w] + rng.integers(-6, 7, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (360, 640), (1080, 1920)])
def test_integral_exact(ctx, oracle, shape):
    img = np.random.default_rng(1).integers(0, 256, shape).astype(np.uint8)
    got = ctx.integral(img)
    assert np.array_equal(got, oracle.integral(img))
    assert got[-1, -1] == int(img.astype(np.int64).sum())


def test_integral_wider_than_one_pass_and_ragged(oracle):
    """More than 2048 sum columns (two column passes of a strip's workgroup), heights that are not multiples of the strip's 8 rows,
    widths that are / are not multiples of 8 (vector / byte pixel loads)."""
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 2600, 70, 256)
    try:
        for shape in ((37, 2500), (64, 2048), (9, 2047), (1, 1), (8, 8), (70, 2600), (17, 2056)):
            img = np.random.default_rng(shape[0] * 7 + shape[1]).integers(0, 256, shape).astype(np.uint8)
            assert np.array_equal(c.integral(img), oracle.integral(img)), shape
    finally:
        c.close()


@pytest.mark.parametrize("octave,layer", [(0, 0), (0, 4), (1, 2), (2, 1), (3, 0), (3, 4)])
def test_hessian_layer_debug_hook_bit_exact(ctx, oracle, octave, layer):
    """uvo_hessian_layer is the parity hook (k_hessian_layer_debug: calcLayerDetAndTrace from the run-time box patterns, the
    arithmetic k_hessian_finish uses); the fused detection kernels k_hessian_nms_c / _p are covered by the keypoint equality of
    the SURF tests below, which any wrong determinant near a maximum breaks."""
    img = _rand_img(2, 360, 640)
    s = ctx.integral(img)
    det, tr = ctx.hessian_layer(img.shape, octave, layer)
    size = (9 + 6 * layer) << octave
    odet, otr = oracle.surf_layer(oracle.integral(img), size, 1 << octave)
    assert np.array_equal(det.view(np.uint32), odet.view(np.uint32))
    assert np.array_equal(tr.view(np.uint32), otr.view(np.uint32))


def _assert_kps_equal(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        av, bv = a[f], b[f]
        if av.dtype.kind == "f":
            assert np.array_equal(av.view(np.uint32), bv.view(np.uint32)), f
        else:
            assert np.array_equal(av, bv), f


@pytest.mark.parametrize("shape,thr", [((120, 160), 50), ((360, 640), 400), ((240, 320), 1500), ((720, 1280), 1500),
                                       ((100, 1920), 300), ((1080, 96), 300), ((67, 125), 30), ((187, 249), 100)])
def test_surf_detect_describe_bit_exact(ctx, oracle, shape, thr):
    import ergo_uvo_amd as uvo
    img = _rand_img(3, *shape)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=thr))
    kps, desc = ctx.detect_features(img)
    okps, odesc = oracle.surf(img, thr)
    assert len(okps) > 20
    _assert_kps_equal(kps, okps)
    assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))


@pytest.mark.parametrize("n_octaves", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(48, 48), (36, 300), (240, 320)])
def test_surf_octave_counts_and_tiny_images_bit_exact(ctx, oracle, shape, n_octaves):
    """nOctaves 1..4 (four takes the launch that carries octaves 0, 2 and 3 together, fewer the per-octave kernels) on images so
    small that the upper octaves have few or no samples: same keypoints and descriptors as the oracle, or none on both sides."""
    import ergo_uvo_amd as uvo
    img = _rand_img(5, *shape)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=20, SURF_OCTAVES_NUMBER=n_octaves))
    try:
        kps, desc = ctx.detect_features(img)
        okps, odesc = oracle.surf(img, 20, n_octaves=n_octaves)
        _assert_kps_equal(kps, okps)
        assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))
        if shape == (240, 320):
            assert len(okps) > 100 and (n_octaves < 3 or int(okps["octave"].max()) >= 2)
    finally:
        ctx.set_params(uvo.Params.stereo())


@pytest.mark.parametrize("shape,thr", [((200, 262), 1), ((96, 350), 20)])
def test_surf_dense_maxima_bit_exact(ctx, oracle, shape, thr):
    """White noise and an almost-zero threshold: every local maximum of every layer is a candidate, so the list of samples
    whose outer-layer determinants are evaluated lazily (k_hessian_finish) is as long as it gets; odd widths take the
    descriptor kernels' scalar paths."""
    import ergo_uvo_amd as uvo
    img = np.random.default_rng(11).integers(0, 256, shape).astype(np.uint8)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=thr))
    kps, desc = ctx.detect_features(img)
    okps, odesc = oracle.surf(img, thr)
    assert len(okps) > 200
    _assert_kps_equal(kps, okps)
    assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))


@pytest.mark.parametrize("shape,thr,upright,extended", [((240, 320), 300, False, False), ((360, 640), 800, True, True), ((360, 640), 800, False, True),
                                                      ((187, 249), 100, False, False), ((1080, 1920), 9000, False, False)])
def test_surf_orientation_and_extended_bit_exact(ctx, oracle, shape, thr, upright, extended):
    """orientation assignment + rotated sampling window (SURF_UPRIGHT = false) and the 128-element descriptor
    (SURF_EXTENDED = true) of detect_features (VO_utility.cpp:114-119 passes both flags to SURF::create)."""
    import ergo_uvo_amd as uvo
    if shape[0] >= 1080:
        from ergo_uvo_amd import synth
        img = synth.mono_frame(synth.Scene(9, 1920), 0, 1920, 1080)              # blobs of every scale: windows up to ~700 px
    else:
        img = _rand_img(3, *shape)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=thr, SURF_UPRIGHT=int(upright), SURF_EXTENDED=int(extended)))
    try:
        kps, desc = ctx.detect_features(img)
        okps, odesc = oracle.surf(img, thr, upright=upright, extended=extended)
        assert len(okps) > 20 and desc.shape[1] == (128 if extended else 64)
        _assert_kps_equal(kps, okps)
        assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))
        if not upright:
            assert len(np.unique(kps["angle"])) > 10
    finally:
        ctx.set_params(uvo.Params.stereo())


def test_surf_capacity_overflow_is_an_error():
    """More candidates than max_kpts (and more NMS survivors than the 4 x max_kpts list) must fail loudly, never truncate."""
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1), 0, 640, 360, 64)
    try:
        with pytest.raises(uvo.UvoError) as ei:
            c.detect_features(_rand_img(3, 360, 640))
        assert ei.value.status == 3          # UVO_CAPACITY
    finally:
        c.close()


def test_surf_empty_and_tiny(ctx, oracle):
    import ergo_uvo_amd as uvo
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=100))
    flat = np.full((64, 64), 77, np.uint8)
    kps, desc = ctx.detect_features(flat)
    assert len(kps) == 0 and desc.shape == (0, 64)
    img = _rand_img(5, 40, 56)          # most layers do not fit
    kps, desc = ctx.detect_features(img)
    okps, odesc = oracle.surf(img, 100)
    _assert_kps_equal(kps, okps)
    assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))


def test_match_knn_and_ratio_bit_exact(ctx, oracle, scene_small):
    L, R = scene_small[0]
    _, dL = oracle.surf(L, 500)
    _, dR = oracle.surf(R, 500)
    idx, dist = ctx.knn_match(dL, dR)
    oidx, odist = oracle.knn2(dL, dR)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    for ratio in (0.7, 0.8):
        m = ctx.match_features(dL, dR, ratio)
        om = oracle.match(dL, dR, ratio)
        assert len(om) > 50
        assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
        assert np.array_equal(m["distance"].view(np.uint32), om["distance"].view(np.uint32))


@pytest.mark.parametrize("seed,kind", [(0, "unit"), (1, "unit"), (2, "clustered"), (3, "scaled"), (4, "tiny")])
def test_match_shortlist_margin_holds(ctx, oracle, seed, kind):
    """The matrix-core shortlist works on bf16 hi/lo splits of the descriptors; whatever it lets through is re-evaluated in
    OpenCV's order, so the result must equal the brute force exactly as long as the margin covers the split's error.
    Random unit vectors put almost every row at the same distance (a dense band around the second-nearest), `clustered` adds
    near-duplicates 1e-4 apart, `scaled` / `tiny` move the magnitudes away from SURF's unit norm."""
    rng = np.random.default_rng(100 + seed)
    nq, nt = 700, 900
    if kind == "clustered":
        base = np.abs(rng.normal(size=(60, 64)))
        dt = base[rng.integers(0, 60, nt)] + rng.normal(scale=1e-4, size=(nt, 64))
        dq = base[rng.integers(0, 60, nq)] + rng.normal(scale=1e-4, size=(nq, 64))
    else:
        dt = rng.normal(size=(nt, 64)); dq = rng.normal(size=(nq, 64))
    dt /= np.linalg.norm(dt, axis=1, keepdims=True); dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    scale = {"scaled": 37.5, "tiny": 3e-3}.get(kind, 1.0)
    dt = (dt * scale).astype(np.float32); dq = (dq * scale).astype(np.float32)
    idx, dist = ctx.knn_match(dq, dt)
    oidx, odist = oracle.knn2(dq, dt)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32))


def test_match_ties_and_edges(ctx, oracle):
    rng = np.random.default_rng(7)
    d2 = rng.normal(size=(300, 64)).astype(np.float32)
    d2[17] = d2[5]; d2[200] = d2[5]; d2[250] = d2[100]          # exact duplicates -> distance ties
    d1 = np.concatenate([d2[[5, 100, 250]], rng.normal(size=(70, 64)).astype(np.float32)])
    idx, dist = ctx.knn_match(d1, d2)
    oidx, odist = oracle.knn2(d1, d2)
    assert np.array_equal(idx, oidx) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    assert list(idx[0]) == [5, 17]                               # ties keep the lower train index first
    # train set with a single row: no second neighbour -> no matches (reference would index out of bounds)
    assert len(ctx.match_features(d1, d2[:1], 0.8)) == 0
    # ... whatever an earlier call left in the staging buffer right behind that row: near (d2[1] == the queries' own rows) or far
    for filler in (np.concatenate([d1[:1], d1[:64]]), np.concatenate([d1[:1], 1e3 * d1[:64]])):
        ctx.knn_match(d1, filler)                                # stages `filler` where d2[:1] goes next
        idx1, dist1 = ctx.knn_match(d1, d2[:1])
        assert np.all(idx1[:, 0] == 0) and np.all(idx1[:, 1] == -1)
        assert np.array_equal(dist1[:, 0].view(np.uint32), oracle.knn2(d1, d2[:1])[1][:, 0].view(np.uint32))
        assert len(ctx.match_features(d1, d2[:1], 0.99)) == 0
    assert len(ctx.match_features(d1[:0], d2, 0.8)) == 0
    # append semantics (VOU:538)
    first = ctx.match_features(d1, d2, 0.9)
    both = ctx.match_features(d1, d2, 0.9, matches=first)
    assert len(both) == 2 * len(first) and np.array_equal(both[:len(first)], first)


def _synthetic_stereo_points(n, seed, noise=0.3):
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(1280)
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    def proj(K, R, t, X):
        Y = X @ R.T + t
        return (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    x1 = proj(rig.K_left, np.eye(3), np.zeros(3), X) + rng.normal(0, noise, (n, 2))
    x2 = proj(rig.K_right, rig.R_right, rig.t_right, X) + rng.normal(0, noise, (n, 2))
    P1 = rig.K_left @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = rig.K_right @ np.hstack([rig.R_right, rig.t_right[:, None]])
    return rig, X, x1.astype(np.float32), x2.astype(np.float32), P1, P2


@pytest.mark.parametrize("n", [1, 7, 500, 3000])
def test_triangulate_bit_exact(ctx, oracle, n):
    rig, X, x1, x2, P1, P2 = _synthetic_stereo_points(n, 11)
    got = ctx.triangulatePoints(P1, P2, x1, x2)
    want = oracle.triangulate(P1, P2, x1, x2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    Xh = (got[:3] / got[3]).T
    assert np.abs(Xh - X).max() < 0.5


@pytest.mark.parametrize("n,noise", [(4, 0.2), (6, 0.2), (800, 0.2), (800, 3.0), (2500, 1.0)])
def test_extract_3dpoints_bit_exact(ctx, oracle, n, noise):
    rig, X, x1, x2, P1, P2 = _synthetic_stereo_points(n, 13, noise)
    p4 = oracle.triangulate(P1, P2, x1, x2)
    p4[:, ::17] *= -1.0 if n > 100 else 1.0          # some points behind the camera / flipped sign
    args = (x1, x2, np.eye(3), np.zeros(3), rig.R_right, rig.t_right, rig.K_left, rig.K_right, p4)
    pts, idx = ctx.extract_3Dpoints(*args)
    opts, oidx = oracle.extract_3d_points(*args, min_pts=5, tol=3.0)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(pts.view(np.uint64), opts.view(np.uint64))


def test_extract_3dpoints_tree_sums_and_ordered_sums_agree(ctx, oracle):
    """extract_3Dpoints' +-3 sigma filter (MU:35-56): the kernel takes the sum and the sum of squares as tree sums, bounds their distance
    from the reference's ordered sums, and lets the ordered chains decide only when some depth lies within that radius of a
    threshold (pose.hip, k_extract3d_b).  Both routes (UVO_EXTRACT3D_SEQ forces the ordered one) and the oracle must keep the same
    points -- also when all depths are equal (zero variance: the radius test cannot vouch for the tree sums), and for 6 to 3000 points."""
    import os
    rig, X, x1, x2, P1, P2 = _synthetic_stereo_points(3000, 17, 0.5)
    p4_full = oracle.triangulate(P1, P2, x1, x2)
    cases = []
    for n in (6, 64, 1025, 3000):
        cases.append((n, p4_full[:, :n].copy(), "plain"))
    cases.append((500, p4_full[:, :500].copy(), "plain"))
    flat = p4_full[:, :300].copy(); flat[2] = flat[3] * 4.0       # every depth 4.0 (in float): zero variance
    cases.append((300, flat, "flat"))
    try:
        for n, p4, tag in cases:
            args = (x1[:n], x2[:n], np.eye(3), np.zeros(3), rig.R_right, rig.t_right, rig.K_left, rig.K_right, p4)
            want_pts, want_idx = oracle.extract_3d_points(*args, min_pts=5, tol=1e9 if tag == "flat" else 3.0)
            for seq in ("0", "1"):
                os.environ["UVO_EXTRACT3D_SEQ"] = seq
                if tag == "flat":
                    import ergo_uvo_amd as uvo
                    ctx.set_params(uvo.Params.stereo(REPROJECTION_TOLERANCE=1e9))
                pts, idx = ctx.extract_3Dpoints(*args)
                if tag == "flat":
                    ctx.set_params(uvo.Params.stereo())
                assert np.array_equal(idx, want_idx), (tag, n, seq)
                assert np.array_equal(pts.view(np.uint64), want_pts.view(np.uint64)), (tag, n, seq)
    finally:
        os.environ.pop("UVO_EXTRACT3D_SEQ", None)


@pytest.mark.parametrize("n", [1, 257, 3000])
def test_reproject_errors_bit_exact(ctx, oracle, n):
    rig, X, x1, x2, P1, P2 = _synthetic_stereo_points(n, 29, 0.7)
    err = ctx.reproject_errors(X, rig.R_right, rig.t_right, rig.K_right, x2)
    oerr = oracle.reproject_errors(X, rig.R_right, rig.t_right, rig.K_right, x2)
    assert np.array_equal(err.view(np.uint64), oerr.view(np.uint64))


@pytest.mark.parametrize("n,outliers,noise", [(5, 0.0, 0.1), (6, 0.0, 0.1), (300, 0.2, 0.3), (1200, 0.5, 0.5), (900, 0.9, 0.3), (40, 1.0, 0.0)])
def test_pnp_ransac_parity(ctx, oracle, n, outliers, noise):
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(1920)
    rng = np.random.default_rng(17 + n)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
    Rt, tt = synth.true_relative_motion()
    Y = X @ Rt.T + tt
    K = rig.K_left
    x = (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    x[bad] = rng.uniform(0, 1000, (int(bad.sum()), 2))
    x = x.astype(np.float32)
    ok, rvec, tvec, inl = ctx.solvePnPRansac(X, x, K, 1000, 1.0, 0.99)
    ook, orvec, otvec, oinl = oracle.solve_pnp_ransac(X, x, K, 1000, 1.0, 0.99)
    assert ok == ook
    assert np.array_equal(inl, oinl)                    # bit-exact inlier set
    def rel(a, b):
        return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
    # the pose comes from the refit on the (bit-exact) inlier set.  With fewer than 24 inliers the refit is the sequential,
    # OpenCV-ordered one (bitwise equal to the oracle); otherwise the workgroup-parallel one of uvo_epnp_fast.h, held to the
    # north_star tolerance of 1e-4 relative (observed: ~1e-11)
    assert rel(rvec, orvec) < 1e-4 and rel(tvec, otvec) < 1e-4
    if len(oinl) < 24:
        assert np.array_equal(rvec.view(np.uint64), orvec.view(np.uint64)), (rvec, orvec)
        assert np.array_equal(tvec.view(np.uint64), otvec.view(np.uint64))
    elif ok:
        assert rel(rvec, orvec) < 1e-8 and rel(tvec, otvec) < 1e-8, (rel(rvec, orvec), rel(tvec, otvec))
    if outliers <= 0.5 and n > 5:
        assert ok and rel(tvec, tt) < 0.05


def test_stereo_sequence_parity(ctx, oracle, scene_small):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500))
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    for k, (L, R) in enumerate(scene_small):
        r = ctx.stereo_step(L, R, 0.05)
        o = ovo.step(L, R, 0.05)
        for f in ("valid", "initialized", "n_left", "n_right", "n_stereo_matches", "n_tri_matches", "n_good3d", "n_inliers"):
            assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
        _assert_kps_equal(ctx.stereo_get("kps_left"), ovo.get("kps_left"))
        _assert_kps_equal(ctx.stereo_get("kps_right"), ovo.get("kps_right"))
        assert np.array_equal(ctx.stereo_get("desc_left").view(np.uint32), ovo.get("desc_left").view(np.uint32))
        ms, oms = ctx.stereo_get("matches_stereo"), ovo.get("matches_stereo")
        assert np.array_equal(ms["queryIdx"], oms["queryIdx"]) and np.array_equal(ms["trainIdx"], oms["trainIdx"])
        if k > 0:
            mt, omt = ctx.stereo_get("matches_tri"), ovo.get("matches_tri")
            assert np.array_equal(mt["queryIdx"], omt["queryIdx"]) and np.array_equal(mt["trainIdx"], omt["trainIdx"])
            assert np.array_equal(ctx.stereo_get("points4d").view(np.uint32), ovo.get("points4d").view(np.uint32))
            assert np.array_equal(ctx.stereo_get("good_idx"), ovo.get("good_idx"))
            assert np.array_equal(ctx.stereo_get("inliers"), ovo.get("inliers"))
            assert r.valid == 1
            for a, b in ((r.rvec, o.rvec), (r.tvec, o.tvec), (r.t_prev_curr, o.t_prev_curr), (r.velocity, o.velocity)):
                a, b = np.array(list(a)), np.array(list(b))
                assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b)          # north_star tolerance (refit: uvo_epnp_fast.h)
    Rt, tt = synth.true_relative_motion()
    assert np.linalg.norm(np.array(list(r.tvec)) - tt) < 0.01


def test_pipelined_submit_collect_equals_step(ctx, scene_small):
    """uvo_stereo_submit/collect (1..4 pairs in flight on separate lanes) must give exactly the
    results of the synchronous uvo_stereo_step."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500))
    seq = [scene_small[k] for k in (0, 1, 2, 1, 0, 1, 2)]

    def fields(r):
        return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers,
                tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr), tuple(r.velocity))

    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    sync = [fields(ctx.stereo_step(L, R, 0.05)) for L, R in seq]
    assert sum(f[0] for f in sync) == len(seq) - 1
    for depth in (2, 1, 4, 3):
        ctx.stereo_set_depth(depth)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)      # resets the VO state
        piped, sub = [], 0
        ctx.stereo_submit(*seq[0]); sub += 1
        piped.append(fields(ctx.stereo_collect(0.05)))                              # the init pair cannot be pipelined
        while len(piped) < len(seq):
            while sub < len(seq) and sub - len(piped) < depth:
                ctx.stereo_submit(*seq[sub]); sub += 1
            if depth > 1 and sub < len(seq):
                with pytest.raises(uvo.UvoError):
                    ctx.stereo_submit(*seq[sub])                                    # pipeline full
            piped.append(fields(ctx.stereo_collect(0.05)))
        assert piped == sync, depth
        with pytest.raises(uvo.UvoError):
            ctx.stereo_collect(0.05)                 # nothing in flight
    ctx.stereo_set_depth(2)


def test_failure_paths_through_the_pipeline(ctx, oracle, scene_small):
    """Frames without features in the middle of a sequence: the gate ladder (VO:556/567/626/634/665), the state kept on
    failure and the empty "after stereo match" sets handed to the next pair must behave as in the reference's loop --
    synchronously, with several pairs in flight, and in the oracle."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500))
    blank = (np.full_like(scene_small[0][0], 90), np.full_like(scene_small[0][1], 90))
    half = (scene_small[1][0], blank[1])                      # features on the left only: VO:556 fails on the right count
    seq = [scene_small[0], scene_small[1], blank, scene_small[2], scene_small[1], half, scene_small[0], scene_small[1], scene_small[2]]

    def fields(r):
        return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers,
                tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr), tuple(r.velocity))

    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    want = [fields(ovo.step(L, R, 0.05)) for L, R in seq]
    assert [w[0] for w in want] == [0, 1, 0, 0, 1, 0, 0, 1, 1]       # init, ok, blank, no prev set, ok, half-blank, no prev set, ok, ok
    def same(got, exp):
        """gate results and counts exactly; poses to the north_star tolerance (the refit is not bitwise: uvo_epnp_fast.h)"""
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            assert g[:8] == e[:8], (g[:8], e[:8])
            for a, b in zip(g[8:], e[8:]):
                a, b = np.array(a), np.array(b)
                assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), (a, b)
        return True

    ctx.stereo_set_depth(1)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    sync = [fields(ctx.stereo_step(L, R, 0.05)) for L, R in seq]
    assert same(sync, want)
    for depth in (2, 4):
        ctx.stereo_set_depth(depth)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        piped, sub = [], 0
        ctx.stereo_submit(*seq[0]); sub += 1
        piped.append(fields(ctx.stereo_collect(0.05)))
        while len(piped) < len(seq):
            while sub < len(seq) and sub - len(piped) < depth:
                ctx.stereo_submit(*seq[sub]); sub += 1
            piped.append(fields(ctx.stereo_collect(0.05)))
        assert piped == sync, depth                 # the pipelined path is the synchronous one, bit for bit
    ctx.stereo_set_depth(2)


# ------------------------------------------------------------------ mono path
def _mono_scene(n, seed, planar=False, noise=0.0, outliers=0.0):
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
    return K, x1.astype(np.float32), x2.astype(np.float32), R, t


def _beq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("n,method,thr,outl", [(5, 8, 1.0, 0.0), (6, 8, 1.0, 0.0), (300, 8, 1.0, 0.25), (300, 4, 0.1, 0.25),
                                               (1500, 8, 0.5, 0.5), (1500, 4, 0.1, 0.4), (40, 8, 1.0, 1.0)])
def test_find_essential_and_recover_pose_parity(ctx, oracle, n, method, thr, outl):
    K, x1, x2, R, t = _mono_scene(n, 100 + n + method, noise=0.15, outliers=outl)
    ok, E, mask = ctx.findEssentialMat(x1, x2, K, method, 0.99, thr, 2000)
    ook, oE, omask = oracle.find_essential_mat(x1, x2, K, method, 0.99, thr, 2000)
    assert ok == ook and np.array_equal(mask, omask)                       # bit-exact inlier mask
    if ok:
        assert np.linalg.norm(E - oE) <= 1e-4 * np.linalg.norm(oE) and _beq(E, oE)
        g, Rr, tr, m2 = ctx.recoverPose(E, x1, x2, K, mask)
        og, oR, ot, om2 = oracle.recover_pose(oE, x1, x2, K, omask)
        assert g == og and np.array_equal(m2, om2) and _beq(Rr, oR) and _beq(tr, ot)


@pytest.mark.parametrize("n,method,thr,outl", [(4, 8, 1.0, 0.0), (5, 8, 1.0, 0.0), (250, 8, 1.0, 0.2), (250, 4, 0.1, 0.2),
                                               (1500, 8, 0.5, 0.5), (1500, 4, 0.1, 0.4), (600, 8, 0.5, 0.7)])
def test_find_homography_parity(ctx, oracle, n, method, thr, outl):
    import ergo_uvo_amd as uvo
    ctx.set_params(uvo.Params.mono())                       # HOMOGRAPHY_DISTANCE = 50 (mono_VO_parameters.yaml:28)
    K, x1, x2, R, t = _mono_scene(n, 200 + n + method, planar=True, noise=0.1, outliers=outl)
    ok, H, mask = ctx.findHomography(x1, x2, method, thr, 2000, 0.99)
    ook, oH, omask = oracle.find_homography(x1, x2, method, thr, 2000, 0.99)
    assert ok == ook and np.array_equal(mask, omask)
    if ok:
        assert np.linalg.norm(H - oH) <= 1e-4 * np.linalg.norm(oH) and _beq(H, oH)      # includes the refit + LM polish
        Rs, ts, ns = ctx.decomposeHomographyMat(H, K)
        oRs, ots, ons = oracle.decompose_homography(oH, K)
        assert _beq(Rs, oRs) and _beq(ts, ots) and _beq(ns, ons)
        g, Rr, tr = ctx.recover_pose_homography(H, x1, x2, K)
        og, oR, ot = oracle.recover_pose_homography(oH, x1, x2, K, 50.0)
        assert g == og
        if g > 0:
            assert _beq(Rr, oR) and _beq(tr, ot)


def test_estimate_relative_pose_parity(ctx, oracle):
    import ergo_uvo_amd as uvo
    for method, planar, start_e in [(8, False, True), (4, False, True), (8, True, False), (4, True, True), (8, False, False)]:
        K, x1, x2, R, t = _mono_scene(400, 300 + method + planar, planar=planar, noise=0.15, outliers=0.2)
        kw = dict(ESSENTIAL_OUTLIER_METHOD=method, HOMOGRAPHY_OUTLIER_METHOD=method)
        if method == 8:
            kw.update(ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0)
        ctx.set_params(uvo.Params.mono(**kw))
        op = oracle.mono_params(method=method)
        if method == 8:
            op.ESSENTIAL_THRESHOLD = 1.0; op.HOMOGRAPHY_THRESHOLD = 1.0
        import ctypes as C
        assert ctx.select_estimation_method(x1, x2) == bool(oracle.lib().orc_select_estimation_method(
            x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), len(x1), 10))
        got = ctx.estimate_relative_pose(x1, x2, K, start_e)
        want = oracle.estimate_relative_pose(op, start_e, x1, x2, K)
        assert got[0] == want[0] and got[1] == want[1]
        assert _beq(got[2], want[2]) and _beq(got[3], want[3])
        assert np.array_equal(got[4], want[4]) and np.array_equal(got[5], want[5]) and np.array_equal(got[6], want[6])


def test_mono_sequence_parity(ctx, oracle, mono_small):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    for method in (8, 4):
        kw = dict(SURF_MIN_HESSIAN=400, ESSENTIAL_OUTLIER_METHOD=method, HOMOGRAPHY_OUTLIER_METHOD=method, REPROJECTION_TOLERANCE=3.0)
        op = oracle.mono_params(400, method=method); op.REPROJECTION_TOLERANCE = 3.0
        if method == 8:
            kw.update(ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0)
            op.ESSENTIAL_THRESHOLD = 1.0; op.HOMOGRAPHY_THRESHOLD = 1.0
        ctx.set_params(uvo.Params.mono(**kw))
        ctx.mono_set_camera(rig.K_left)
        ovo = oracle.MonoVO(op, rig.K_left)
        blank = np.full((360, 640), 90, np.uint8)
        seq = [blank, mono_small[0], mono_small[1], mono_small[2], blank, mono_small[1], mono_small[0]]
        n_pub = 0
        for k, img in enumerate(seq):
            r = ctx.mono_step(img, 4.0, 0.2)
            o = ovo.step(img, 4.0, 0.2)
            for f in ("published", "valid", "initialized", "used_essential", "success", "n_kps", "n_matches", "n_inliers", "n_good3d", "n_front"):
                assert getattr(r, f) == getattr(o, f), (method, k, f, getattr(r, f), getattr(o, f))
            n_pub += r.published
            if r.published:
                for a, b in ((r.R, o.R), (r.t, o.t), (r.velocity, o.velocity), ([r.SF], [o.SF])):
                    a, b = np.array(list(a)), np.array(list(b))
                    assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12)      # north_star tolerance
                    assert _beq(a, b)
                assert np.array_equal(ctx.mono_get("mask"), ovo.get("mask"))
                m, om = ctx.mono_get("matches"), ovo.get("matches")
                assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
                assert _beq(ctx.mono_get("good_pts"), ovo.get("good_pts"))
        assert n_pub >= 3


@pytest.mark.parametrize("depth", [2, 3])
def test_mono_pipelined_submit_collect_equals_step(mono_small, depth):
    """uvo_mono_submit / uvo_mono_collect with `depth` frames in flight give uvo_mono_step's results bit for bit, including the
    frames that fail a gate (blank images) and the state they leave behind (R, t, SF kept)."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    blank = np.full((360, 640), 90, np.uint8)
    seq = [blank, mono_small[0], mono_small[1], mono_small[2], blank, mono_small[1], mono_small[0], mono_small[2], mono_small[1], blank,
           mono_small[0], mono_small[1]]
    fields = ("published", "valid", "initialized", "used_essential", "success", "n_kps", "n_matches", "n_inliers", "n_good3d", "n_front")
    for method in (8, 4):
        kw = dict(SURF_MIN_HESSIAN=400, ESSENTIAL_OUTLIER_METHOD=method, HOMOGRAPHY_OUTLIER_METHOD=method, REPROJECTION_TOLERANCE=3.0)
        if method == 8:
            kw.update(ESSENTIAL_THRESHOLD=1.0, HOMOGRAPHY_THRESHOLD=1.0)
        a = uvo.Context(uvo.Params.mono(**kw), 0, 640, 360, 8192)
        b = uvo.Context(uvo.Params.mono(**kw), 0, 640, 360, 8192)
        try:
            a.mono_set_camera(rig.K_left); b.mono_set_camera(rig.K_left)
            b.stereo_set_depth(depth)
            want = []
            for img in seq:
                r = a.mono_step(img, 4.0, 0.2)
                want.append((r, a.mono_get("mask").copy(), a.mono_get("matches").copy(), a.mono_get("good_pts").copy()))
            got = []
            sub = 0
            for i in range(len(seq)):
                while sub < len(seq) and sub - i < depth:
                    b.mono_submit(seq[sub], 4.0); sub += 1
                r = b.mono_collect(0.2)
                got.append((r, b.mono_get("mask").copy(), b.mono_get("matches").copy(), b.mono_get("good_pts").copy()))
            n_pub = 0
            for k, ((rw, mw, tw, gw), (rg, mg, tg, gg)) in enumerate(zip(want, got)):
                for f in fields:
                    assert getattr(rg, f) == getattr(rw, f), (method, depth, k, f)
                n_pub += rw.published
                if rw.published:
                    for x, y in ((rg.R, rw.R), (rg.t, rw.t), (rg.velocity, rw.velocity), ([rg.SF], [rw.SF])):
                        assert _beq(np.array(list(x)), np.array(list(y))), (method, depth, k)
                    assert np.array_equal(mg, mw) and np.array_equal(tg, tw) and _beq(gg, gw)
            assert n_pub >= 5
            with pytest.raises(uvo.UvoError):
                b.mono_step(seq[1], 4.0, 0.2)                 # mixing needs a reset
            b.mono_reset()
            assert b.mono_step(seq[1], 4.0, 0.2).initialized == 0
        finally:
            a.close(); b.close()


def test_device_inputs_are_ordered_after_the_producing_torch_stream(scene_small):
    """A frame still being written on a torch stream when it is handed over: the uploads wait for that stream (uvo_ctx_set_producer_stream),
    so the result is the one of the finished frame; submitted tensors stay alive until the collect even when the caller drops them."""
    import torch
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 8192)
    try:
        L, R = scene_small[0]
        want_k, want_d = c.detect_features(L)
        side = torch.cuda.Stream()
        src = torch.from_numpy(L).cuda()
        big = torch.randn(4096, 4096, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(20):
                big = big @ big * 1e-4                     # keeps the stream busy for a while
            frame = torch.zeros_like(src)
            frame.copy_(src)                               # the frame appears only after the matmuls
            got_k, got_d = c.detect_features(frame)         # "torch" mode: ordered after `side`, the current stream here
        assert np.array_equal(got_k, want_k) and np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32))
        # lifetime: the tensors of a submitted pair are dropped by the caller before the pair is collected
        c.stereo_set_depth(3)
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        want = [c.stereo_step(*p, 0.05) for p in scene_small]
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        c.stereo_step(*scene_small[0], 0.05)
        for k in (1, 2):
            a, b = torch.from_numpy(scene_small[k][0]).cuda(), torch.from_numpy(scene_small[k][1]).cuda()
            c.stereo_submit(a, b)
            del a, b
            torch.cuda.empty_cache()
        for k in (1, 2):
            r = c.stereo_collect(0.05)
            assert (r.valid, r.n_left, r.n_inliers, tuple(r.tvec)) == (want[k].valid, want[k].n_left, want[k].n_inliers, tuple(want[k].tvec))
    finally:
        c.close()


def test_misuse_is_refused_loudly(scene_small):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    with pytest.raises(uvo.UvoError):                        # solvePnPRansac flags other than EPNP are not silently replaced
        uvo.Context(uvo.Params.stereo(PNP_METHOD_FLAG=2), 0, 640, 360, 1024)
    with pytest.raises(uvo.UvoError):
        uvo.Context(uvo.Params.stereo(USE_EXTRINSIC_GUESS=1), 0, 640, 360, 1024)
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, 8192)
    try:
        assert c.warning == ""                               # two lanes by default: the default hardware queues are enough
        with pytest.raises(uvo.UvoError):
            c.set_params(uvo.Params.stereo(PNP_METHOD_FLAG=0))
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        c.stereo_step(*scene_small[0], 0.05)
        c.stereo_submit(*scene_small[1])
        d = np.zeros((8, 64), np.float32)
        for call in (lambda: c.detect_features(scene_small[0][0]), lambda: c.knn_match(d, d), lambda: c.integral(scene_small[0][0]),
                     lambda: c.triangulatePoints(np.eye(3, 4), np.eye(3, 4), np.zeros((2, 2)), np.zeros((2, 2))),
                     lambda: c.mono_collect(0.05)):
            with pytest.raises(uvo.UvoError):
                call()                                       # lane 0's buffers belong to the pair in flight; a stereo pair is not a mono frame
        assert c.stereo_collect(0.05).valid == 1
        c.detect_features(scene_small[0][0])
        with pytest.raises(uvo.UvoError):
            c.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, SURF_EXTENDED=1))   # the running loop holds 64-element rows of the previous pair
        c.stereo_reset()
        c.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, SURF_EXTENDED=1))
        assert c.detect_features(scene_small[0][0])[1].shape[1] == 128
        with pytest.raises(ValueError):
            c.knn_match(d, d)                                # 64-element rows into a 128-element context
    finally:
        c.close()


# ---------------------------------------------------------------- SURF_EXTENDED / SURF_UPRIGHT = 0 through the matcher and the loops
@pytest.mark.parametrize("seed,kind", [(0, "unit"), (2, "clustered"), (4, "tiny")])
def test_match_128_element_rows_bit_exact(oracle, seed, kind):
    """BFMatcher(NORM_L2).knnMatch(k = 2) + ratio on 128-element rows (SURF extended): indices and distances bitwise, on random rows
    with near-duplicates (the row-by-row scan path) and on real extended descriptors."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rng = np.random.default_rng(seed)
    c = uvo.Context(uvo.Params.stereo(SURF_EXTENDED=1), 0, 640, 480, 4096)
    try:
        for n1, n2 in ((700, 900), (1, 1), (130, 1), (257, 129), (3000, 3000) if kind == "unit" else (300, 260)):
            a = rng.standard_normal((n1, 128)).astype(np.float32)
            b = rng.standard_normal((n2, 128)).astype(np.float32)
            if kind == "clustered" and n2 > 8:
                b[n2 // 2:] = b[: n2 - n2 // 2] + rng.standard_normal((n2 - n2 // 2, 128)).astype(np.float32) * 1e-4
                a[: min(n1, n2) // 3] = b[: min(n1, n2) // 3]
            if kind == "tiny":
                a *= 1e-3; b *= 1e-3
            a /= np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12) if kind != "tiny" else 1.0
            idx, dist = c.knn_match(a, b)
            oidx, odist = oracle.knn2(a, b)
            assert np.array_equal(idx, oidx), (kind, n1, n2)
            assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32)), (kind, n1, n2)
            m, om = c.match_features(a, b, 0.8), oracle.match(a, b, 0.8)
            assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
        scene = synth.Scene(11, 640)
        L, R = synth.stereo_pair(scene, 0, 640, 480)
        _, dL = oracle.surf(L, 500, extended=True)
        _, dR = oracle.surf(R, 500, extended=True)
        m, om = c.match_features(dL, dR, 0.7), oracle.match(dL, dR, 0.7)
        assert len(om) > 50 and np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
        assert np.array_equal(m["distance"].view(np.uint32), om["distance"].view(np.uint32))
    finally:
        c.close()


@pytest.mark.parametrize("extended,upright", [(1, 1), (0, 0), (1, 0)])
def test_stereo_and_mono_loops_with_extended_and_oriented_surf(oracle, scene_small, mono_small, extended, upright):
    """The whole stereo and mono steps with every SURF switch of the parameter files (surf_extended, surf_upright): counts, both match
    lists, inlier sets bitwise, pose to 1e-4, synchronous and with three pairs in flight."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, SURF_EXTENDED=extended, SURF_UPRIGHT=upright), 0, 640, 360, 8192)
    try:
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        op = oracle.stereo_params(1500); op.SURF_EXTENDED = extended; op.SURF_UPRIGHT = upright
        ovo = oracle.StereoVO(op, rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        seq = [scene_small[k] for k in (0, 1, 2, 1, 0)]
        ref = []
        for k, (L, R) in enumerate(seq):
            r, o = c.stereo_step(L, R, 0.05), ovo.step(L, R, 0.05)
            for f in ("valid", "initialized", "n_left", "n_right", "n_stereo_matches", "n_tri_matches", "n_good3d", "n_inliers"):
                assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
            assert np.array_equal(c.stereo_get("desc_left").view(np.uint32), ovo.get("desc_left").view(np.uint32))
            assert c.stereo_get("desc_left").shape[1] == (128 if extended else 64)
            ms, oms = c.stereo_get("matches_stereo"), ovo.get("matches_stereo")
            assert np.array_equal(ms["queryIdx"], oms["queryIdx"]) and np.array_equal(ms["trainIdx"], oms["trainIdx"])
            if k > 0:
                mt, omt = c.stereo_get("matches_tri"), ovo.get("matches_tri")
                assert np.array_equal(mt["queryIdx"], omt["queryIdx"]) and np.array_equal(mt["trainIdx"], omt["trainIdx"])
                assert np.array_equal(c.stereo_get("inliers"), ovo.get("inliers"))
                assert r.valid == 1
                for a, b in ((r.rvec, o.rvec), (r.tvec, o.tvec), (r.t_prev_curr, o.t_prev_curr)):
                    a, b = np.array(list(a)), np.array(list(b))
                    assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b)
            ref.append((r.valid, r.n_left, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers, tuple(r.tvec)))
        # the same sequence with three pairs in flight
        c.stereo_reset(); c.stereo_set_depth(3)
        got, pending = [], 0
        for L, R in seq:
            if pending == 3:
                got.append(c.stereo_collect(0.05)); pending -= 1
            c.stereo_submit(L, R); pending += 1
        while pending:
            got.append(c.stereo_collect(0.05)); pending -= 1
        for r, e in zip(got, ref):
            assert (r.valid, r.n_left, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers) == e[:6]
            assert np.linalg.norm(np.array(list(r.tvec)) - np.array(e[6])) <= 1e-4 * max(np.linalg.norm(np.array(e[6])), 1e-12)
    finally:
        c.close()
    kw = dict(SURF_MIN_HESSIAN=400, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8, REPROJECTION_TOLERANCE=3.0, ESSENTIAL_THRESHOLD=1.0,
              HOMOGRAPHY_THRESHOLD=1.0, SURF_EXTENDED=extended, SURF_UPRIGHT=upright)
    op = oracle.mono_params(400, method=8); op.REPROJECTION_TOLERANCE = 3.0; op.ESSENTIAL_THRESHOLD = 1.0; op.HOMOGRAPHY_THRESHOLD = 1.0
    op.SURF_EXTENDED = extended; op.SURF_UPRIGHT = upright
    c = uvo.Context(uvo.Params.mono(**kw), 0, 640, 360, 8192)
    try:
        c.mono_set_camera(rig.K_left)
        ovo = oracle.MonoVO(op, rig.K_left)
        n_pub = 0
        for k, img in enumerate([mono_small[0], mono_small[1], mono_small[2], mono_small[1]]):
            r, o = c.mono_step(img, 4.0, 0.2), ovo.step(img, 4.0, 0.2)
            for f in ("published", "valid", "initialized", "used_essential", "success", "n_kps", "n_matches", "n_inliers", "n_good3d", "n_front"):
                assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
            n_pub += r.published
            if r.published:
                assert np.array_equal(c.mono_get("mask"), ovo.get("mask"))
                m, om = c.mono_get("matches"), ovo.get("matches")
                assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
                for a, b in ((r.R, o.R), (r.t, o.t)):
                    a, b = np.array(list(a)), np.array(list(b))
                    assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12)
        assert n_pub >= 2
    finally:
        c.close()


def test_contexts_come_and_go_without_leaking_device_memory(scene_small):
    """Twelve contexts with six lanes each, used and destroyed: the device's free memory returns to where it was."""
    import torch
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    torch.cuda.synchronize()
    free0 = None
    for it in range(12):
        c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 1920, 1080, 8192)
        c.stereo_set_depth(6)
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        for L, R in scene_small:
            c.stereo_submit(L, R)
        for _ in scene_small:
            c.stereo_collect(0.05)
        c.stereo_submit(*scene_small[0])                # destroyed with a pair still in flight
        c.close()
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        if it == 1:
            free0 = free                                # after the first rounds: allocator pools and code objects are in place
        if it > 1:
            assert free0 - free < 64 << 20, (it, free0, free)


def test_a_later_context_runs_at_the_first_ones_rate():
    """a context created after another one of the same process had been destroyed ran ~10 % below its rate
    (the runtime binds a stream to a hardware queue once; which stream serves which role decides which roles share a pipe of the
    command processor: tools/probe/ctx_reuse.py).  The library parks its streams per role, so the fifth context of a process gets the
    first one's binding -- pinned here at 5 % (the probe measures 0.99 .. 1.01; create / destroy gave 0.91 .. 0.94, one shared free
    list 0.87 on every second context).  Measured by the probe itself in a process of its own: inside the test runner the HIP runtime
    is already up with its default of four hardware queues, the loop is then bound by the host (3000 pairs/s against 4700) and what
    the comparison sees is the speed of the Python thread, not the streams' binding."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("UVO_STREAM_POOL",)}
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "probe", "ctx_reuse.py"), "240"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"contexts 1\.\.5: ([0-9 ]+?) +5th/1st", p.stdout)
    assert m, p.stdout[-500:]
    rates = [float(x) for x in m.group(1).split()]
    assert len(rates) == 5 and min(rates[1:]) >= 0.95 * rates[0], rates


@pytest.mark.parametrize("W,H", [(641, 363), (644, 360), (1000, 562)])
def test_stereo_loop_at_odd_geometries_from_device_images(oracle, W, H):
    """Widths that are not multiples of 8 / 4 (byte-wise integral loads, scalar descriptor taps) and images handed over as device
    tensors (read in place when their base is 16-byte aligned; a misaligned view is copied): the loop still equals the oracle's."""
    import torch
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    scene = synth.Scene(77, W)
    rig = synth.stereo_rig(W)
    pairs = [synth.stereo_pair(scene, k, W, H) for k in range(3)]
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1200), 0, W, H, 8192)
    try:
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ovo = oracle.StereoVO(oracle.stereo_params(1200), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        flat = torch.empty(2 * W * H + 64, dtype=torch.uint8, device="cuda")
        for k, (L, R) in enumerate(pairs):
            off = 0 if k != 1 else 3                                   # pair 1 from a view that is not 16-byte aligned
            dl = flat[off: off + W * H].view(H, W); dr = flat[off + W * H + 16: off + 2 * W * H + 16].view(H, W)
            dl.copy_(torch.from_numpy(L)); dr.copy_(torch.from_numpy(R))
            torch.cuda.synchronize()
            r, o = c.stereo_step(dl, dr, 0.05), ovo.step(L, R, 0.05)
            for f in ("valid", "initialized", "n_left", "n_right", "n_stereo_matches", "n_tri_matches", "n_good3d", "n_inliers"):
                assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
            _assert_kps_equal(c.stereo_get("kps_left"), ovo.get("kps_left"))
            assert np.array_equal(c.stereo_get("desc_right").view(np.uint32), ovo.get("desc_right").view(np.uint32))
            if k > 0:
                assert r.valid == 1 and np.array_equal(c.stereo_get("inliers"), ovo.get("inliers"))
    finally:
        c.close()


@pytest.mark.parametrize("bytes_", [32, 61, 64, 7])
def test_hamming_branch_of_match_features_bit_exact(ctx, oracle, bytes_):
    """VO_utility.cpp:520-524 (AKAZE / ORB descriptors): BFMatcher(NORM_HAMMING).knnMatch(k = 2) + ratio.  Integer distances tie all
    the time on random bits, so this is mostly a test of the insertion order (lower train index first) across 512-row chunks."""
    rng = np.random.default_rng(bytes_)
    for n1, n2 in ((900, 1300), (1, 1), (257, 1), (3000, 2999), (40, 513)):
        a = rng.integers(0, 256, (n1, bytes_), dtype=np.uint8)
        b = rng.integers(0, 256, (n2, bytes_), dtype=np.uint8)
        if n2 > 600:
            b[550:560] = b[10:20]                       # exact duplicates on both sides of a chunk boundary
            a[:10] = b[10:20]                           # distance 0 twice: indices 10.. and 550.. must come out in that order
        idx, dist = ctx.knn_match_hamming(a, b)
        oidx, odist = oracle.knn2_hamming(a, b)
        assert np.array_equal(idx, oidx), (bytes_, n1, n2)
        assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
        for ratio in (0.8, 0.97):
            m, om = ctx.match_features_hamming(a, b, ratio), oracle.match_hamming(a, b, ratio)
            assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
            assert np.array_equal(m["distance"].view(np.uint32), om["distance"].view(np.uint32))


def test_sift_arm_of_the_l2_branch_bit_exact(ctx, oracle):
    """VO_utility.cpp:525-529: "SIFT" descriptors go to the same BFMatcher(NORM_L2) as SURF's; they are 128 floats per row whatever
    SURF_EXTENDED says (this context's own rows are 64 wide).  SIFT rows are small non-negative integers stored as floats, so equal
    distances -- ties -- are common: the order of insertion (lower train index first) is what is compared, and the distances bitwise."""
    rng = np.random.default_rng(128)

    def rows(n, base=None):
        a = np.abs(rng.normal(size=(n, 128))) ** 2 if base is None else np.clip(base + rng.normal(size=base.shape) * 5.0, 0, None)
        a = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-9) * 512.0
        return np.clip(np.rint(a), 0, 255).astype(np.float32)

    for n1, n2 in ((1200, 1500), (1, 2), (130, 129), (3000, 2950)):
        b = rows(n2)
        a = rows(n1)
        if n2 > 600:
            a[:400] = rows(400, base=b[100:500].astype(np.float64))
            b[520:530] = b[100:110]                     # duplicated train rows across a 128-row chunk boundary
        idx, dist = ctx.knn_match(a, b, dim=128)
        oidx, odist = oracle.knn2(a, b)
        assert np.array_equal(idx, oidx), (n1, n2)
        assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
        for ratio in (0.7, 0.8):
            m, om = ctx.match_features(a, b, ratio, dim=128), oracle.match(a, b, ratio)
            assert np.array_equal(m["queryIdx"], om["queryIdx"]) and np.array_equal(m["trainIdx"], om["trainIdx"])
            assert np.array_equal(m["distance"].view(np.uint32), om["distance"].view(np.uint32))
    with pytest.raises(Exception):
        ctx.knn_match(np.zeros((4, 96), np.float32), np.zeros((4, 96), np.float32), dim=96)


@pytest.mark.parametrize("kw", [dict(REPROJECTION_ERROR_THRESHOLD=0.03), dict(ITERATIONS_COUNT=30), dict(REPROJECTION_ERROR_THRESHOLD=0.03, ITERATIONS_COUNT=40),
                                dict(MIN_NUM_3DPOINTS=100000)])
def test_speculative_pnp_round_and_its_fallbacks(ctx, oracle, scene_small, kw):
    """uvo_stereo_step queues the first RANSAC round on the device without the host (pose.hip: k_pnp_*_spec) and accepts it only when
    the host's own replay of the scan agrees; otherwise the stage runs again the worker-driven way.  Cases that must fall back
    or change the round: a 0.03-px threshold (few inliers: the adaptive count reaches past 64 hypotheses), fewer than 64
    iterations allowed, both, and a 3-D point gate that never opens.  Every case must give the oracle's inlier sets bit for bit
    synchronously (speculative path) AND with pairs in flight (worker path), and the two must agree bitwise."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, **kw))
    op = oracle.stereo_params(1500)
    for k, v in kw.items():
        setattr(op, k, v)
    seq = [scene_small[0], scene_small[1], scene_small[2], scene_small[1], scene_small[0]]
    ovo = oracle.StereoVO(op, rig.K_left, rig.K_right, rig.R_right, rig.t_right)

    def fields(r):
        return (r.valid, r.n_good3d, r.n_inliers, tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr))

    try:
        ctx.stereo_set_depth(1)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        sync = []
        for L, R in seq:
            r = ctx.stereo_step(L, R, 0.05)
            o = ovo.step(L, R, 0.05)
            assert (r.valid, r.n_good3d, r.n_inliers) == (o.valid, o.n_good3d, o.n_inliers), kw
            assert np.array_equal(ctx.stereo_get("inliers"), ovo.get("inliers")), kw
            for a, b in ((r.rvec, o.rvec), (r.tvec, o.tvec)):
                a, b = np.array(list(a)), np.array(list(b))
                assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-300) or not o.n_inliers, kw
            sync.append(fields(r))
        ctx.stereo_set_depth(3)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        piped, sub = [], 0
        ctx.stereo_submit(*seq[0]); sub += 1
        piped.append(fields(ctx.stereo_collect(0.05)))
        while len(piped) < len(seq):
            while sub < len(seq) and sub - len(piped) < 3:
                ctx.stereo_submit(*seq[sub]); sub += 1
            piped.append(fields(ctx.stereo_collect(0.05)))
        assert piped == sync, kw
    finally:
        ovo.close()
        ctx.stereo_set_depth(2)
        ctx.set_params(uvo.Params.stereo())
