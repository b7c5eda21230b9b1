"""Known-answer tests for the mono-path restatements in oracle/ (five-point essential matrix,
RANSAC / LMedS registrators, recoverPose, homography DLT + LM, decomposeHomographyMat, the
uvo_libraries mono functions and the mono_VO loop).  PARITY vs OpenCV is UNPINNED; these pin the oracle
against geometry with known ground truth and against numpy."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

K = np.array([[800.0, 0, 320.0], [0, 790.0, 240.0], [0, 0, 1.0]])


def _scene(n, seed, planar=False, noise=0.0, outliers=0.0, R=None, t=None):
    rng = np.random.default_rng(seed)
    R = Rotation.from_rotvec([0.02, -0.03, 0.015]).as_matrix() if R is None else R
    t = np.array([0.30, -0.08, 0.12]) if t is None else t      # depth/baseline < 50: recoverPose drops farther points
    X = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(3.0, 7.0, n)], 1)
    if planar:
        nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
        X[:, 2] = (5.0 - X[:, 0] * nrm[0] - X[:, 1] * nrm[1]) / nrm[2]          # plane n.X = 5
    def proj(Y):
        return (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    x1 = proj(X) + rng.normal(0, noise, (n, 2))
    x2 = proj(X @ R.T + t) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    x2[bad] = rng.uniform(0, 600, (int(bad.sum()), 2))
    return X, x1.astype(np.float32), x2.astype(np.float32), R, t, bad


def test_solve_poly_and_jacobi_eigen_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    for deg in (3, 6, 10):
        roots = rng.normal(size=deg)
        c = np.poly(roots)[::-1]                       # increasing powers
        got = oracle.solve_poly(c)
        assert np.allclose(np.sort(got.real), np.sort(roots), atol=1e-6) and np.abs(got.imag).max() < 1e-6
    c = np.array([5.0, 0, 1.0])                        # z^2 + 5: purely imaginary pair
    got = oracle.solve_poly(c)
    assert np.allclose(np.sort(got.imag), [-np.sqrt(5), np.sqrt(5)]) and np.abs(got.real).max() < 1e-12
    for n in (3, 8, 9):
        A = rng.normal(size=(n, n)); A = A @ A.T
        w, v = oracle.jacobi_eigen(A)
        assert np.allclose(w, np.linalg.eigvalsh(A)[::-1], rtol=1e-10, atol=1e-10) and np.all(np.diff(w) <= 1e-12)
        assert np.allclose(v @ A @ v.T, np.diag(w), atol=1e-9) and np.allclose(v @ v.T, np.eye(n), atol=1e-12)


def _essential(R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E)


def test_five_point_contains_true_essential(oracle):
    X, x1, x2, R, t, _ = _scene(5, 3)
    Kinv = np.linalg.inv(K)
    q1 = (np.hstack([x1.astype(np.float64), np.ones((5, 1))]) @ Kinv.T)[:, :2]
    q2 = (np.hstack([x2.astype(np.float64), np.ones((5, 1))]) @ Kinv.T)[:, :2]
    Es = oracle.five_point(q1, q2)
    assert 1 <= len(Es) <= 10
    Et = _essential(R, t)
    errs = [min(np.linalg.norm(E - Et), np.linalg.norm(E + Et)) for E in Es]
    assert min(errs) < 1e-4                              # float32 pixel coordinates limit the accuracy
    for E in Es:                                         # every returned model satisfies the five constraints + unit norm
        res = [np.array([*q2[i], 1]) @ E @ np.array([*q1[i], 1]) for i in range(5)]
        assert np.abs(res).max() < 1e-9 and abs(np.linalg.norm(E) - 1) < 1e-12
        s = np.linalg.svd(E, compute_uv=False)
        assert abs(s[0] - s[1]) < 1e-6 and s[2] < 1e-6   # essential-matrix constraint


@pytest.mark.parametrize("method", [8, 4])
def test_find_essential_and_recover_pose(oracle, method):
    X, x1, x2, R, t, bad = _scene(300, 5, noise=0.15, outliers=0.25)
    ok, E, mask = oracle.find_essential_mat(x1, x2, K, method, 0.99, 0.1 if method == 4 else 1.0, 2000)
    assert ok
    assert mask[bad].mean() < 0.1 and mask[~bad].mean() > 0.6
    g, Rr, tr, m2 = oracle.recover_pose(E, x1, x2, K, mask)
    assert g == int(m2.sum()) and np.all(m2 <= mask) and g > 0.5 * (~bad).sum()
    # findEssentialMat returns the raw 5-point model of the best sample (no refit), so the pose carries the
    # noise of five correspondences
    assert np.allclose(Rr, R, atol=2e-2)
    assert np.linalg.norm(tr - t / np.linalg.norm(t)) < 0.15 and abs(np.linalg.norm(tr) - 1) < 1e-12
    # Sampson error of the true E on clean points is ~0
    import ctypes as C
    Kinv = np.linalg.inv(K)
    Xc, c1, c2, *_ = _scene(50, 6)
    q1 = np.ascontiguousarray((np.hstack([c1.astype(np.float64), np.ones((50, 1))]) @ Kinv.T)[:, :2])
    q2 = np.ascontiguousarray((np.hstack([c2.astype(np.float64), np.ones((50, 1))]) @ Kinv.T)[:, :2])
    err = np.empty(50, np.float32); Et = np.ascontiguousarray(_essential(R, t))
    oracle.lib().orc_sampson_error(q1.ctypes.data_as(C.c_void_p), q2.ctypes.data_as(C.c_void_p), 50, Et.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p))
    assert err.max() < 1e-9


def test_homography_dlt_lm_and_decomposition(oracle):
    X, x1, x2, R, t, bad = _scene(250, 7, planar=True, noise=0.1, outliers=0.2)
    nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm); d = 5.0
    Ht = K @ (R + np.outer(t, nrm) / d) @ np.linalg.inv(K); Ht /= Ht[2, 2]
    for method, thr in ((8, 1.0), (4, 0.1)):
        ok, H, mask = oracle.find_homography(x1, x2, method, thr, 2000, 0.99)
        assert ok and H[2, 2] == 1.0
        assert np.abs(H - Ht).max() / np.abs(Ht).max() < 5e-3
        assert mask[bad].mean() < 0.1 and mask[~bad].mean() > 0.55
    Rs, ts, ns = oracle.decompose_homography(Ht, K)
    assert len(Rs) == 4
    best = min(range(4), key=lambda i: np.linalg.norm(Rs[i] - R) + np.linalg.norm(ts[i] - t / d) + np.linalg.norm(ns[i] - nrm))
    assert np.allclose(Rs[best], R, atol=1e-3) and np.allclose(ts[best], t / d, atol=1e-3) and np.allclose(ns[best], nrm, atol=1e-3)
    for i in range(4):
        assert np.allclose(Rs[i] @ Rs[i].T, np.eye(3), atol=1e-6) and np.linalg.det(Rs[i]) > 0
        Hi = Rs[i] + np.outer(ts[i], ns[i])              # every candidate reproduces the (normalised) homography
        Hn = np.linalg.inv(K) @ Ht @ K
        Hn = Hn / np.linalg.svd(Hn, compute_uv=False)[1]
        assert min(np.abs(Hi - Hn).max(), np.abs(Hi + Hn).max()) < 1e-3
    # pure rotation: a single solution with zero translation
    Hrot = K @ R @ np.linalg.inv(K)
    Rs, ts, ns = oracle.decompose_homography(Hrot, K)
    assert len(Rs) == 1 and np.allclose(Rs[0], R, atol=1e-9) and not ts.any()
    # minimal kernel on exact data
    import ctypes as C
    Xp, p1, p2, *_ = _scene(4, 8, planar=True)
    H4 = np.empty((3, 3))
    assert oracle.lib().orc_homography_kernel(p1.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p), 4, H4.ctypes.data_as(C.c_void_p)) == 1
    proj = np.hstack([p1.astype(np.float64), np.ones((4, 1))]) @ H4.T
    assert np.abs(proj[:, :2] / proj[:, 2:] - p2).max() < 1e-3
    # checkSubset: three collinear points are rejected
    col = np.array([[0, 0], [1, 1], [5, 3], [2, 2]], np.float32)
    assert oracle.lib().orc_homography_check_subset(col.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p), 4) == 0
    assert oracle.lib().orc_homography_check_subset(p1.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p), 4) == 1


def test_mono_library_functions(oracle):
    import ctypes as C
    lib = oracle.lib()
    X, x1, x2, R, t, bad = _scene(200, 11, noise=0.1, outliers=0.2)
    # select_estimation_method: median displacement against the int threshold DISTANCE (VOU:739)
    disp = np.sqrt(((x1.astype(np.float64) - x2.astype(np.float64)) ** 2).sum(1))
    med = np.median(disp)
    assert lib.orc_select_estimation_method(x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), 200, int(med) + 1) == 0
    assert lib.orc_select_estimation_method(x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), 200, int(med)) == 1
    # estimate_relative_pose: essential branch succeeds on a general scene
    p = oracle.mono_params(method=4)
    ok, used_e, Rr, tr, in1, in2, mask = oracle.estimate_relative_pose(p, True, x1, x2, K)
    assert ok and used_e and np.allclose(Rr, R, atol=2e-2) and len(in1) == len(in2) >= int(mask.sum())
    # ... and switches to the other method once when the first one fails the inlier-fraction test (VOU:165-178)
    p2 = oracle.mono_params(method=8)
    p2.VPF_THRESHOLD = 2.0                          # unreachable: both methods fail
    ok, used_e, *_ = oracle.estimate_relative_pose(p2, True, x1, x2, K)
    assert not ok and not used_e                    # flipped exactly once
    # convert_3Dpoints_camera keeps the ORIGINAL rows whose transformed z is positive (VOU:55-57)
    pts = np.array([[0, 0, 5.0], [0, 0, 0.5], [1, 1, 2.0]])
    Rz = np.eye(3); tz = np.array([0, 0, -1.0])
    out = np.empty((3, 3))
    k = lib.orc_convert_3Dpoints_camera(pts.ctypes.data_as(C.c_void_p), 3, Rz.ctypes.data_as(C.c_void_p), tz.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert k == 2 and np.array_equal(out[:2], pts[[0, 2]])
    # compute_scale_factor = (float)range / median(z)
    lib.orc_compute_scale_factor.restype = C.c_double
    sf = lib.orc_compute_scale_factor(C.c_float(4.1), pts.ctypes.data_as(C.c_void_p), 3)
    assert sf == float(np.float32(4.1)) / 2.0
    assert lib.orc_compute_scale_factor(C.c_float(4.1), pts.ctypes.data_as(C.c_void_p), 0) == 0.0


def test_recover_pose_homography_type_punning(oracle):
    """VOU:601-602 reads a double out of two neighbouring floats; with it the count depends on z_{j+1}'s bits."""
    X, x1, x2, R, t, _ = _scene(120, 13, planar=True)
    nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
    Ht = K @ (R + np.outer(t, nrm) / 5.0) @ np.linalg.inv(K); Ht /= Ht[2, 2]
    g, Rr, tr = oracle.recover_pose_homography(Ht, x1, x2, K, 50.0)
    assert 0 <= g <= 119                              # the last column is never counted (read past the buffer in the reference)
    if g > 0:
        assert abs(np.linalg.norm(tr) - 1) < 1e-12 and np.allclose(Rr @ Rr.T, np.eye(3), atol=1e-6)
    # emulate the reinterpretation independently for one candidate: depth z as float, pairs (z_j, z_{j+1})
    z = np.array([1.5, 2.5, 3.0, 4.0, -1.0, 0.5], np.float32)
    vals = np.frombuffer(np.stack([z[:-1], z[1:]], 1).tobytes(), np.float64)
    good = int(((vals > 0) & (vals < 50.0)).sum())
    assert good == 3                                  # high words 2.5, 3.0 and 0.5 -> small positive doubles; 4.0 -> 512; -1.0 -> negative


def test_mono_sequence(oracle, mono_small):
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    p = oracle.mono_params(400, method=8)
    p.ESSENTIAL_THRESHOLD = 1.0; p.HOMOGRAPHY_THRESHOLD = 1.0; p.REPROJECTION_TOLERANCE = 3.0
    vo = oracle.MonoVO(p, rig.K_left)
    blank = np.full((360, 640), 90, np.uint8)
    r = vo.step(blank)                                # VO:240: too few features, not initialised
    assert r.initialized == 0 and r.published == 0
    r = vo.step(mono_small[0])
    assert r.initialized == 0 and r.n_kps > 100       # this frame initialises
    R4, C4 = synth.camera_pose(4)                      # X_4 = R4 (X_0 - C4)
    Rt, tt = R4, -R4 @ C4
    r = vo.step(mono_small[1], 4.0, 0.2)
    assert r.published == 1 and r.success == 1 and r.valid == 1 and r.used_essential == 1
    assert np.allclose(np.array(list(r.R)).reshape(3, 3), Rt, atol=2e-2)
    tdir = np.array(list(r.t)); assert np.linalg.norm(tdir - tt / np.linalg.norm(tt)) < 0.4   # unrefined 5-point model
    assert r.n_good3d >= 5 and r.n_front >= 5 and r.SF > 0
    v = np.array(list(r.velocity)); R = np.array(list(r.R)).reshape(3, 3)
    assert np.allclose(v, (-r.SF) * (1.0 / 0.2) * (R.T @ tdir), rtol=1e-12)
    # scale: range / median(z) with |t| = 1 -> SF ~ |true translation| when range ~ true median depth
    assert 0.5 * np.linalg.norm(tt) < r.SF < 2.0 * np.linalg.norm(tt)
    r2 = vo.step(blank, 4.0, 0.2)                     # VO:276-284: skipped frame publishes nothing and rolls the state
    assert r2.published == 0
    r3 = vo.step(mono_small[2], 4.0, 0.2)             # prev is now the blank frame: no matches -> skipped again
    assert r3.published == 0 and r3.n_matches == 0
    r4 = vo.step(mono_small[1], 4.0, 0.2)             # backwards motion 8 -> 4
    assert r4.published == 1 and r4.success == 1
