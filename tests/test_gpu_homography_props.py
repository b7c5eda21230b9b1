"""Checks of the product's HOST-side homography code that share no text with oracle/o_homography.c ("break the
twin"): `mono.hip`'s Levenberg-Marquardt refinement inside uvo_find_homography (cv::findHomography, VO_utility.cpp:152) and
uvo_decompose_homography_mat (cv::decomposeHomographyMat, VO_utility.cpp:585) were restated by one author in two files, so
HIP-vs-oracle agreement proves nothing for them.  What is asserted here follows from the DEFINITION of the two operations and is
evaluated in numpy from the values the C ABI returns:

  * the refined H is a stationary point of the reprojection cost over its own inlier set (the gradient J^T r of the 8-parameter
    cost vanishes relative to the residual) and is no worse than a direct-linear-transform fit of the same inliers;
  * every (R, t, n) of a decomposition satisfies R in SO(3), |n| = 1 and K (R + t n^T) K^-1 proportional to H -- to 1e-6, not to
    1e-9: cv::decomposeHomographyMat (HomographyDecompInria::decompose) takes `v = 2 * sqrtf(1 + trace(S) - M00 - M11 - M22)` with the
    FLOAT square root, and R = H (I - (2 / v) t* n^T) inherits its 6e-8 relative error; the restatement keeps that `sqrtf` (observed
    here: orthogonality defect 5e-8 median, 2.8e-7 worst over 400 solutions; a sign, index or transposition mistake is O(1)).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _project(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:3]


def _residuals_and_jacobian(H, p1, p2):
    """r = proj(H, p1) - p2 (2n) and dr/dh for h = the first eight entries of H with H[2][2] held fixed (the parametrisation a
    refinement of cv::findHomography's result uses); written from the projective map's derivative, not from any solver's code."""
    x, y = p1[:, 0], p1[:, 1]
    w = H[2, 0] * x + H[2, 1] * y + H[2, 2]
    u = (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / w
    v = (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / w
    n = len(p1)
    J = np.zeros((2 * n, 8))
    J[0::2, 0] = x / w; J[0::2, 1] = y / w; J[0::2, 2] = 1 / w
    J[1::2, 3] = x / w; J[1::2, 4] = y / w; J[1::2, 5] = 1 / w
    J[0::2, 6] = -u * x / w; J[0::2, 7] = -u * y / w
    J[1::2, 6] = -v * x / w; J[1::2, 7] = -v * y / w
    r = np.empty(2 * n)
    r[0::2] = u - p2[:, 0]; r[1::2] = v - p2[:, 1]
    return r, J


def _dlt(p1, p2):
    """Hartley-normalised direct linear transform (numpy SVD): an independent estimate whose cost the refined H must not exceed."""
    def norm(p):
        c = p.mean(0); s = np.sqrt(2) / np.mean(np.linalg.norm(p - c, axis=1))
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        return (p - c) * s, T
    a, Ta = norm(p1); b, Tb = norm(p2)
    A = []
    for (x, y), (u, v) in zip(a, b):
        A.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u]); A.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    h = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
    Hn = np.linalg.inv(Tb) @ h @ Ta
    return Hn / Hn[2, 2]


@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.mono(), 0, 640, 480, 2048)
    yield c
    c.close()


@pytest.mark.parametrize("method", [8, 4])          # RANSAC, LMEDS
def test_refined_homography_is_a_stationary_point_of_its_inliers_cost(ctx, method):
    worst_grad, worst_gain = 0.0, 0.0
    for seed in range(50):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(40, 400))
        # a plane seen from two views: H = K (R + t n^T / d) K^-1 with a few degrees of rotation, plus pixel noise and gross outliers
        K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
        ang = rng.normal(0, 0.05, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
        Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
        Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
        t = rng.normal(0, 0.3, 3)
        nrm = np.array([0.0, 0.0, 1.0])
        Htrue = K @ (Rx @ Ry @ Rz + np.outer(t, nrm) / 5.0) @ np.linalg.inv(K)
        p1 = np.c_[rng.uniform(20, 620, n), rng.uniform(20, 460, n)]
        p2 = _project(Htrue, p1) + rng.normal(0, float(rng.choice([0.05, 0.3, 0.8])), (n, 2))
        n_out = int(n * rng.choice([0.0, 0.1, 0.3]))
        if n_out:
            p2[:n_out] = np.c_[rng.uniform(0, 640, n_out), rng.uniform(0, 480, n_out)]
        p1f, p2f = p1.astype(np.float32), p2.astype(np.float32)
        ok, H, mask = ctx.findHomography(p1f, p2f, method=method, threshold=3.0, max_iters=2000, confidence=0.995)
        assert ok and abs(H[2, 2] - 1.0) < 1e-12, (seed, ok, H)
        inl = mask.astype(bool)
        assert inl.sum() >= max(8, int(0.5 * (n - n_out))), (seed, int(inl.sum()), n, n_out)
        a, b = p1f[inl].astype(np.float64), p2f[inl].astype(np.float64)
        r, J = _residuals_and_jacobian(H, a, b)
        g = J.T @ r
        # stationarity, scale-free: the gradient against what a residual of this size could produce through this Jacobian
        rel = np.abs(g).max() / (np.linalg.norm(J, axis=0).max() * np.linalg.norm(r) + 1e-300)
        worst_grad = max(worst_grad, rel)
        assert rel < 1e-6, (seed, method, rel)
        cost, cost_dlt = float(r @ r), float(np.sum((_project(_dlt(a, b), a) - b) ** 2))
        worst_gain = max(worst_gain, cost / cost_dlt)
        assert cost <= cost_dlt * (1 + 1e-9), (seed, method, cost, cost_dlt)
        # and it is a minimum, not merely a stationary point: small steps along every parameter do not lower the cost
        for k in range(8):
            for sgn in (-1.0, 1.0):
                Hp = H.copy().reshape(-1); Hp[k] += sgn * 1e-4 * max(abs(Hp[k]), 1e-3); Hp = Hp.reshape(3, 3)
                rp, _ = _residuals_and_jacobian(Hp, a, b)
                assert float(rp @ rp) >= cost * (1 - 1e-9), (seed, k, sgn)
    print(f"method {method}: worst relative gradient {worst_grad:.3g}, worst cost / DLT cost {worst_gain:.6f}")


def test_every_homography_decomposition_reconstructs_its_homography():
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import _lib
    lib = _lib.lib()
    import ctypes as C
    worst = 0.0
    n_total = 0
    for seed in range(100):
        rng = np.random.default_rng(7000 + seed)
        f = float(rng.uniform(300, 1500))
        K = np.array([[f, float(rng.normal(0, 0.5)), float(rng.uniform(200, 1000))], [0, f * float(rng.uniform(0.9, 1.1)), float(rng.uniform(150, 600))], [0, 0, 1]])
        if seed % 2 == 0:                                   # a genuine plane-induced homography, arbitrary scale and sign
            A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A); Q *= np.sign(np.linalg.det(Q))
            ang = float(rng.uniform(0.01, 0.6)); ax = Q[:, 0]
            Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            t = rng.normal(0, 0.4, 3)
            H = K @ (R + np.outer(t, nrm)) @ np.linalg.inv(K) * float(rng.choice([-1, 1]) * rng.uniform(0.2, 5))
        else:                                               # any well-conditioned 3 x 3 matrix near a similarity
            H = K @ (np.eye(3) + rng.normal(0, 0.15, (3, 3))) @ np.linalg.inv(K) * float(rng.uniform(0.5, 2))
        Rs = np.empty((4, 3, 3)); ts = np.empty((4, 3)); ns = np.empty((4, 3)); n = C.c_int(0)
        Hc, Kc = np.ascontiguousarray(H), np.ascontiguousarray(K)
        st = lib.uvo_decompose_homography_mat(Hc.ctypes.data_as(C.c_void_p), Kc.ctypes.data_as(C.c_void_p), Rs.ctypes.data_as(C.c_void_p),
                                              ts.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), C.byref(n))
        assert st == 0 and n.value in (1, 2, 4), (seed, st, n.value)
        Hn = np.linalg.inv(K) @ H @ K                       # the Euclidean homography, up to scale
        Hn = Hn / np.linalg.svd(Hn, compute_uv=False)[1]    # its middle singular value is 1 for R + t n^T
        for k in range(n.value):
            R, t, nv = Rs[k], ts[k], ns[k]
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and abs(np.linalg.det(R) - 1) < 1e-6, (seed, k)
            assert abs(np.linalg.norm(nv) - 1) < 1e-12, (seed, k, nv)
            G = R + np.outer(t, nv)
            err = min(np.abs(G - Hn).max(), np.abs(G + Hn).max())          # proportional: the sign of H is not observable
            worst = max(worst, err)
            assert err < 1e-6 * max(1.0, np.abs(Hn).max()), (seed, k, err)
            n_total += 1
    assert n_total >= 300
    print(f"{n_total} decompositions, worst |R + t n^T -+ H| = {worst:.3g}")
