"""numpy (float64) statements of three operators of the hot path, written from their published definitions -- NOT from `oracle/`
(nothing here imports it): the second statement tests/test_gpu_definitions.py holds the C ABI's outputs against.

  * surf64_upright   the upright 64-float SURF descriptor (Bay et al., section 4.2, in the form of opencv_contrib's SURFInvoker that the
                     reference reaches through detect_features, VO_utility.cpp:114-119): a square window of side floor(21 s),
                     s = size * 1.2 / 9, centred on the keypoint and clamped to the image; area-averaged to 21 x 21 grey levels (8-bit,
                     round to nearest); 20 x 20 Haar differences weighted by a sigma = 3.3 Gaussian; 4 x 4 cells of 5 x 5 samples, each
                     (sum dx, sum dy, sum |dx|, sum |dy|); unit length.
  * cv_rng_stream / ransac_subsets   cv::RNG's multiply-with-carry generator and RANSAC's subset draws (ptsetreg.cpp getSubset).
  * homography_dlt   the normalised 4-point (or n-point) direct linear transform.
  * pose_polish      maximum-likelihood polish of a PnP pose (scipy least_squares on the reprojection residuals)."""
import numpy as np


def gaussian_kernel(n=20, sigma=3.3):
    x = np.arange(n) - (n - 1) * 0.5
    g = np.exp(-x * x / (2.0 * sigma * sigma))
    return g / g.sum()


def area_weights(src: int, dst: int) -> np.ndarray:
    """[dst, src] weights of the area average: destination cell k covers [k * src / dst, (k + 1) * src / dst) of the source axis."""
    scale = src / dst
    W = np.zeros((dst, src))
    for k in range(dst):
        a, b = k * scale, (k + 1) * scale
        for x in range(int(np.floor(a)), min(int(np.ceil(b)), src)):
            W[k, x] = max(0.0, min(b, x + 1) - max(a, x))
    return W / W.sum(1, keepdims=True)


def round_half_even(v):
    return np.rint(v)


def surf_window(img: np.ndarray, x: float, y: float, size: float):
    """The descriptor window of an upright keypoint in the window's own axes: first index walks +x of the image, second index walks
    -y (the upright descriptor's fixed direction is 270 degrees), integer sample positions clamped to the image.  None when the
    wavelets do not fit the image at all."""
    h, w = img.shape
    s = np.float32(size) * np.float32(1.2) / np.float32(9.0)
    win = int(np.float32(21.0) * s)
    if win < 1:
        return None, win
    off = -np.float32(win - 1) / np.float32(2)
    sx = int(np.rint(np.float32(x) + off))
    sy = int(np.rint(np.float32(y) - off))
    xs = np.clip(sx + np.arange(win), 0, w - 1)
    ys = np.clip(sy - np.arange(win), 0, h - 1)
    return img[np.ix_(ys, xs)].T.astype(np.float64), win            # [i: +x][j: -y]


def surf64_upright(img: np.ndarray, x: float, y: float, size: float) -> np.ndarray:
    win, n = surf_window(img, x, y, size)
    if win is None:
        return None
    W = area_weights(n, 21)
    patch = np.clip(round_half_even(W @ win @ W.T), 0, 255)          # 21 x 21 grey levels, 8-bit
    g = gaussian_kernel()
    dw = np.outer(g, g)
    p00, p01, p10, p11 = patch[:-1, :-1], patch[:-1, 1:], patch[1:, :-1], patch[1:, 1:]
    vx = (p01 - p00 + p11 - p10) * dw                                # difference along the second index
    vy = (p10 - p00 + p11 - p01) * dw                                # difference along the first index
    out = []
    for ci in range(4):
        for cj in range(4):
            a, b = vx[ci * 5:ci * 5 + 5, cj * 5:cj * 5 + 5], vy[ci * 5:ci * 5 + 5, cj * 5:cj * 5 + 5]
            out += [a.sum(), b.sum(), np.abs(a).sum(), np.abs(b).sum()]
    out = np.array(out)
    return out / (np.sqrt((out * out).sum()) + np.finfo(np.float32).eps)


# ---------------------------------------------------------------------------------------------- cv::RNG, RANSAC subsets
def cv_rng_stream(seed=0xFFFFFFFFFFFFFFFF):
    """cv::RNG: state = (uint32)state * 4164903690 + (state >> 32); the output is the low 32 bits of the new state."""
    state = seed
    while True:
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        yield state & 0xFFFFFFFF


def ransac_subsets(count: int, m: int, n_subsets: int):
    """The index subsets RANSAC draws, in order: for each, m draws uniform(0, count) = next() % count, a draw equal to an earlier index of
    the same subset is drawn again.  (A subset the estimator's own check refuses still consumed its draws: the list is a superset of the
    subsets a model is fitted to, in stream order.)"""
    rng = cv_rng_stream()
    out = []
    for _ in range(n_subsets):
        sub = []
        while len(sub) < m:
            v = next(rng) % count
            if v not in sub:
                sub.append(v)
        out.append(sub)
    return out


def homography_dlt(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """q ~ H p by the normalised direct linear transform (Hartley): both point sets translated to their centroid and scaled to mean
    absolute deviation 1 per axis, the 9-vector of the smallest singular value of the 2n x 9 system, de-normalised, H[2, 2] = 1."""
    def norm(x):
        c = x.mean(0)
        s = np.abs(x - c).mean(0)
        s = np.where(s > 1e-12, 1.0 / s, 1.0)
        T = np.array([[s[0], 0, -c[0] * s[0]], [0, s[1], -c[1] * s[1]], [0, 0, 1]])
        return (x - c) * s, T
    pn, Tp = norm(p.astype(np.float64))
    qn, Tq = norm(q.astype(np.float64))
    A = []
    for (x, y), (X, Y) in zip(pn, qn):
        A.append([x, y, 1, 0, 0, 0, -x * X, -y * X, -X])
        A.append([0, 0, 0, x, y, 1, -x * Y, -y * Y, -Y])
    h = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
    H = np.linalg.inv(Tq) @ h @ Tp
    return H / H[2, 2] if abs(H[2, 2]) > 1e-300 else H


def homography_err2(H: np.ndarray, p: np.ndarray, q: np.ndarray) -> np.ndarray:
    ph = np.c_[p.astype(np.float64), np.ones(len(p))] @ H.T
    return ((ph[:, :2] / ph[:, 2:] - q.astype(np.float64)) ** 2).sum(1)


# ---------------------------------------------------------------------------------------------- PnP
def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-300:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def project(X, rvec, t, K):
    Y = np.asarray(X, np.float64) @ rodrigues(rvec).T + np.asarray(t, np.float64)
    return (Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])


def reprojection_rms(X, x, rvec, t, K):
    return float(np.sqrt(((project(X, rvec, t, K) - x) ** 2).sum(1).mean()))


def pose_polish(X, x, rvec, t, K):
    from scipy.optimize import least_squares
    f = lambda v: (project(X, v[:3], v[3:], K) - x).ravel()
    r = least_squares(f, np.r_[rvec, t], method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    return r.x[:3], r.x[3:]
