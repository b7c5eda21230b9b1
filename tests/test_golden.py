"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).
They are self-generated -- the reference has no fixtures and OpenCV is absent, so PARITY vs OpenCV is
UNPINNED -- and serve two purposes: the oracle must keep reproducing them bit for bit (CPU tests), and
the HIP path must reproduce them through the C ABI (gpu tests) without the oracle in the loop."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


def _kps_equal(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        assert np.array_equal(_bits(a[f]) if a[f].dtype.kind == "f" else a[f], _bits(b[f]) if b[f].dtype.kind == "f" else b[f]), f


# ------------------------------------------------------------------ oracle vs fixtures (CPU)
def test_oracle_reproduces_surf_fixture(oracle):
    g = _load("surf_320x180.npz")
    s = oracle.integral(g["img"])
    assert int(s.astype(np.int64).sum()) == int(g["integral_crc"][0]) and s[-1, -1] == g["integral_last"][0]
    det, tr = oracle.surf_layer(s, 15, 1)
    assert np.array_equal(_bits(det[60:64]), _bits(g["det15_rows"])) and np.array_equal(_bits(tr[60:64]), _bits(g["trace15_rows"]))
    kps, desc = oracle.surf(g["img"], 800)
    _kps_equal(kps, g["kps"])
    assert np.array_equal(_bits(desc), _bits(g["desc"])) and len(kps) > 100


def test_oracle_reproduces_match_pnp_tri_fixtures(oracle):
    g = _load("match_120x150.npz")
    idx, dist = oracle.knn2(g["d1"], g["d2"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(_bits(dist), _bits(g["dist"]))
    m = oracle.match(g["d1"], g["d2"], 0.8)
    assert np.array_equal(m["queryIdx"], g["matches"]["queryIdx"]) and np.array_equal(m["trainIdx"], g["matches"]["trainIdx"])
    g = _load("pnp_60.npz")
    ok, rvec, tvec, inl = oracle.solve_pnp_ransac(g["X"], g["x"], g["K"], 1000, 1.0, 0.99)
    assert ok == bool(g["ok"][0]) and np.array_equal(inl, g["inliers"])
    assert np.array_equal(_bits(rvec), _bits(g["rvec"])) and np.array_equal(_bits(tvec), _bits(g["tvec"]))
    g = _load("tri_40.npz")
    p4 = oracle.triangulate(g["P1"], g["P2"], g["x1"], g["x2"])
    assert np.array_equal(_bits(p4), _bits(g["points4d"]))
    pts, idx = oracle.extract_3d_points(g["x1"], g["x2"], np.eye(3), np.zeros(3), g["R2"], g["t2"], g["K1"], g["K2"], p4)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(_bits(pts), _bits(g["pts"]))
    assert 0 < len(idx) < 40


# ------------------------------------------------------------------ HIP path vs fixtures (no oracle involved)
@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 640, 480, 8192)
    yield c
    c.close()


@pytest.mark.gpu
def test_hip_reproduces_surf_fixture(ctx):
    import ergo_uvo_amd as uvo
    g = _load("surf_320x180.npz")
    s = ctx.integral(g["img"])
    assert int(s.astype(np.int64).sum()) == int(g["integral_crc"][0])
    det, tr = ctx.hessian_layer(g["img"].shape, 0, 1)
    assert np.array_equal(_bits(det[60:64]), _bits(g["det15_rows"])) and np.array_equal(_bits(tr[60:64]), _bits(g["trace15_rows"]))
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=800))
    kps, desc = ctx.detect_features(g["img"])
    _kps_equal(kps, g["kps"])
    assert np.array_equal(_bits(desc), _bits(g["desc"]))


@pytest.mark.gpu
def test_hip_reproduces_match_pnp_tri_fixtures(ctx):
    g = _load("match_120x150.npz")
    idx, dist = ctx.knn_match(g["d1"], g["d2"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(_bits(dist), _bits(g["dist"]))
    m = ctx.match_features(g["d1"], g["d2"], 0.8)
    assert np.array_equal(m["queryIdx"], g["matches"]["queryIdx"]) and np.array_equal(m["trainIdx"], g["matches"]["trainIdx"])
    assert np.array_equal(_bits(m["distance"]), _bits(g["matches"]["distance"]))
    g = _load("pnp_60.npz")
    ok, rvec, tvec, inl = ctx.solvePnPRansac(g["X"], g["x"], g["K"], 1000, 1.0, 0.99)
    assert ok == bool(g["ok"][0]) and np.array_equal(inl, g["inliers"])
    assert np.linalg.norm(tvec - g["tvec"]) <= 1e-4 * np.linalg.norm(g["tvec"])          # north_star tolerance
    assert np.linalg.norm(rvec - g["rvec"]) <= 1e-4 * np.linalg.norm(g["rvec"])          # (the refit on >= 24 inliers is parallel, not bitwise)
    g = _load("tri_40.npz")
    p4 = ctx.triangulatePoints(g["P1"], g["P2"], g["x1"], g["x2"])
    assert np.array_equal(_bits(p4), _bits(g["points4d"]))
    pts, idx = ctx.extract_3Dpoints(g["x1"], g["x2"], np.eye(3), np.zeros(3), g["R2"], g["t2"], g["K1"], g["K2"], p4)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(_bits(pts), _bits(g["pts"]))


def test_oracle_reproduces_preproc_fixture(oracle):
    g = _load("preproc_96x160.npz")
    dw = int(g["desired_width"][0]); dh = g["out"].shape[0]
    assert np.array_equal(oracle.resize_area_c3(g["rgb"], dw, dh), g["small"])
    assert np.array_equal(oracle.rgb2gray(g["small"]), g["gray"])
    assert np.array_equal(oracle.undistort(g["gray"], g["K"], g["dist"], g["newK"]), g["undistorted"])
    assert np.array_equal(oracle.clahe(g["undistorted"], float(g["clip_limit"][0])), g["out"])
    assert np.array_equal(oracle.get_image(g["rgb"], dw, g["K"], g["dist"], g["newK"], True, int(g["clip_limit"][0])), g["out"])


@pytest.mark.gpu
def test_hip_reproduces_preproc_fixture(ctx):
    g = _load("preproc_96x160.npz")
    dw = int(g["desired_width"][0])
    assert np.array_equal(ctx.get_image(g["rgb"], dw, g["K"], g["dist"], g["newK"], True, int(g["clip_limit"][0])), g["out"])
    assert np.array_equal(ctx.get_image(g["rgb"], dw, g["K"], g["dist"], g["newK"], False, 0), g["undistorted"])
