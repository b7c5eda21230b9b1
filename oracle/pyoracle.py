"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (ergo_uvo_amd) never imports this module.
PARITY UNPINNED vs OpenCV (see oracle/uvo_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


class DMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int), ("trainIdx", C.c_int), ("imgIdx", C.c_int), ("distance", C.c_float)]


class SurfParams(C.Structure):
    _fields_ = [("hessianThreshold", C.c_double), ("nOctaves", C.c_int), ("nOctaveLayers", C.c_int),
                ("extended", C.c_int), ("upright", C.c_int)]


class VoParams(C.Structure):
    _fields_ = [("DISTANCE", C.c_int), ("LOWE_RATIO_THRESHOLD", C.c_double),
                ("ESSENTIAL_OUTLIER_METHOD", C.c_int), ("ESSENTIAL_MAX_ITERS", C.c_double),
                ("ESSENTIAL_CONFIDENCE", C.c_double), ("ESSENTIAL_THRESHOLD", C.c_double),
                ("HOMOGRAPHY_OUTLIER_METHOD", C.c_int), ("HOMOGRAPHY_MAX_ITERS", C.c_double),
                ("HOMOGRAPHY_CONFIDENCE", C.c_double), ("HOMOGRAPHY_THRESHOLD", C.c_double),
                ("HOMOGRAPHY_DISTANCE", C.c_double), ("VPF_THRESHOLD", C.c_double),
                ("REPROJECTION_TOLERANCE", C.c_double), ("MIN_NUM_FEATURES", C.c_int),
                ("MIN_NUM_3DPOINTS", C.c_int), ("MIN_NUM_INLIERS", C.c_int), ("ITERATIONS_COUNT", C.c_int),
                ("REPROJECTION_ERROR_THRESHOLD", C.c_double), ("CONFIDENCE", C.c_double),
                ("USE_EXTRINSIC_GUESS", C.c_int), ("PNP_METHOD_FLAG", C.c_int),
                ("SURF_MIN_HESSIAN", C.c_int), ("SURF_OCTAVES_NUMBER", C.c_int), ("SURF_OCTAVES_LAYERS", C.c_int),
                ("SURF_EXTENDED", C.c_int), ("SURF_UPRIGHT", C.c_int)]


class StereoResult(C.Structure):
    _fields_ = [("valid", C.c_int), ("initialized", C.c_int), ("n_left", C.c_int), ("n_right", C.c_int),
                ("n_stereo_matches", C.c_int), ("n_tri_matches", C.c_int), ("n_good3d", C.c_int),
                ("n_inliers", C.c_int), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3),
                ("t_prev_curr", C.c_double * 3), ("velocity", C.c_double * 3)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])
DM_DTYPE = np.dtype([("queryIdx", "i4"), ("trainIdx", "i4"), ("imgIdx", "i4"), ("distance", "f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # UVO_ORACLE_LIB: another build of the same sources (the sanitizer build of `make -C oracle asan`, tests/test_oracle_sanitizers.py)
        _lib = C.CDLL(os.environ.get("UVO_ORACLE_LIB") or _LIB_PATH)
        _lib.orc_l2_distance_f32.restype = C.c_float
        _lib.orc_hypot.restype = C.c_double
        _lib.orc_hypot.argtypes = [C.c_double, C.c_double]
        _lib.orc_acos.restype = C.c_double
        _lib.orc_acos.argtypes = [C.c_double]
        _lib.orc_sincos.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
        _lib.orc_compute_median.restype = C.c_double
        _lib.orc_stereo_create.restype = C.c_void_p
        _lib.orc_stereo_destroy.argtypes = [C.c_void_p]
        _lib.orc_stereo_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib.orc_stereo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        _lib.orc_rng_init.argtypes = [C.c_void_p, C.c_uint64]
        _lib.orc_rng_next.restype = C.c_uint32
        _lib.orc_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def stereo_params(min_hessian=1500) -> VoParams:
    """stereo_VO_parameters.yaml:20-47."""
    p = VoParams()
    p.LOWE_RATIO_THRESHOLD = 0.8
    p.REPROJECTION_TOLERANCE = 3.0
    p.MIN_NUM_FEATURES = 5
    p.MIN_NUM_3DPOINTS = 5
    p.MIN_NUM_INLIERS = 5
    p.ITERATIONS_COUNT = 1000
    p.REPROJECTION_ERROR_THRESHOLD = 1.0
    p.CONFIDENCE = 0.99
    p.USE_EXTRINSIC_GUESS = 0
    p.PNP_METHOD_FLAG = 1
    p.SURF_MIN_HESSIAN = int(min_hessian)
    p.SURF_OCTAVES_NUMBER = 4
    p.SURF_OCTAVES_LAYERS = 3
    p.SURF_EXTENDED = 0
    p.SURF_UPRIGHT = 1
    return p


# ------------------------------------------------------------------ SURF
def integral(img: np.ndarray) -> np.ndarray:
    img = _c(img, np.uint8)
    h, w = img.shape
    out = np.empty((h + 1, w + 1), np.int32)
    lib().orc_integral_u8(_p(img), w, h, w, _p(out))
    return out


def surf_layer(sum_: np.ndarray, size: int, step: int):
    h, w = sum_.shape[0] - 1, sum_.shape[1] - 1
    det = np.zeros((h // step, w // step), np.float32)
    tr = np.zeros_like(det)
    s = _c(sum_, np.int32)
    lib().orc_surf_layer(_p(s), w, h, size, step, _p(det), _p(tr))
    return det, tr


def surf(img: np.ndarray, hessian=1500.0, n_octaves=4, n_layers=3, cap=20000, extended=False, upright=True):
    img = _c(img, np.uint8)
    h, w = img.shape
    sp = SurfParams(float(hessian), n_octaves, n_layers, int(bool(extended)), int(bool(upright)))
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 128 if extended else 64), np.float32)
    n = lib().orc_surf_detect_and_compute(_p(img), w, h, w, C.byref(sp), _p(kps), _p(desc), cap)
    if n < 0:
        raise RuntimeError(f"oracle SURF: capacity {cap} too small ({-n} keypoints)")
    return kps[:n].copy(), desc[:n].copy()


def resize_area(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = _c(src, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_area_u8(_p(src), src.shape[1], src.shape[0], _p(out), dw, dh)
    return out


# ------------------------------------------------------------------ matching
def knn2(d1: np.ndarray, d2: np.ndarray):
    d1 = _c(d1, np.float32)
    d2 = _c(d2, np.float32)
    idx = np.empty((len(d1), 2), np.int32)
    dist = np.empty((len(d1), 2), np.float32)
    lib().orc_knn2(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1] if d1.ndim == 2 else 64, _p(idx), _p(dist))
    return idx, dist


def match(d1: np.ndarray, d2: np.ndarray, ratio: float) -> np.ndarray:
    d1 = _c(d1, np.float32)
    d2 = _c(d2, np.float32)
    out = np.zeros(max(len(d1), 1), DM_DTYPE)
    m = C.c_int(0)
    lib().orc_match_knn2_ratio(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1] if d1.ndim == 2 else 64, C.c_float(ratio), _p(out), len(out), C.byref(m))
    return out[:m.value].copy()


# ------------------------------------------------------------------ geometry
def knn2_hamming(d1: np.ndarray, d2: np.ndarray):
    """BFMatcher(NORM_HAMMING).knnMatch(k = 2) on uint8 rows (VOU:520-524)."""
    d1 = _c(d1, np.uint8); d2 = _c(d2, np.uint8)
    idx = np.empty((len(d1), 2), np.int32); dist = np.empty((len(d1), 2), np.float32)
    lib().orc_knn2_hamming(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1], _p(idx), _p(dist))
    return idx, dist


def match_hamming(d1: np.ndarray, d2: np.ndarray, ratio: float) -> np.ndarray:
    d1 = _c(d1, np.uint8); d2 = _c(d2, np.uint8)
    out = np.zeros(max(len(d1), 1), DM_DTYPE)
    m = C.c_int(0)
    lib().orc_match_knn2_ratio_hamming(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1], C.c_float(ratio), _p(out), len(out), C.byref(m))
    return out[:m.value].copy()


def triangulate(P1, P2, x1, x2) -> np.ndarray:
    P1 = _c(P1, np.float64); P2 = _c(P2, np.float64)
    x1 = _c(x1, np.float32); x2 = _c(x2, np.float32)
    n = len(x1)
    out = np.empty((4, n), np.float32)
    lib().orc_triangulate_points(_p(P1), _p(P2), _p(x1), _p(x2), n, _p(out))
    return out


def rodrigues_vec2mat(r):
    r = _c(r, np.float64); R = np.empty((3, 3))
    lib().orc_rodrigues_vec2mat(_p(r), _p(R))
    return R


def rodrigues_mat2vec(R):
    R = _c(R, np.float64); r = np.empty(3)
    lib().orc_rodrigues_mat2vec(_p(R), _p(r))
    return r


def extract_3d_points(k1, k2, R1, t1, R2, t2, K1, K2, points4d, min_pts=5, tol=3.0):
    k1 = _c(k1, np.float32); k2 = _c(k2, np.float32)
    n = len(k1)
    p4 = _c(points4d, np.float32)
    pts = np.empty((max(n, 1), 3)); idx = np.empty(max(n, 1), np.int32)
    a = [_c(x, np.float64) for x in (R1, t1, R2, t2, K1, K2)]
    g = lib().orc_extract_3Dpoints(_p(k1), _p(k2), n, *[_p(x) for x in a], _p(p4), min_pts, C.c_double(tol), _p(pts), _p(idx))
    return pts[:g].copy(), idx[:g].copy()


def reproject_errors(world, R, t, K, img):
    world = _c(world, np.float64); img = _c(img, np.float32)
    R, t, K = _c(R, np.float64), _c(t, np.float64), _c(K, np.float64)
    n = len(world)
    err = np.empty(max(n, 1))
    lib().orc_reproject_errors(_p(world), n, _p(R), _p(t), _p(K), _p(img), _p(err))
    return err[:n].copy()


def solve_pnp_ransac(obj, img, K, iters=1000, reproj=1.0, conf=0.99):
    obj = _c(obj, np.float64); img = _c(img, np.float32); K = _c(K, np.float64)
    n = len(obj)
    rvec = np.zeros(3); tvec = np.zeros(3); inl = np.empty(max(n, 1), np.int32); ni = C.c_int(0)
    ok = lib().orc_solve_pnp_ransac(_p(obj), _p(img), n, _p(K), iters, C.c_float(reproj), C.c_double(conf),
                                    _p(rvec), _p(tvec), _p(inl), C.byref(ni))
    return bool(ok), rvec, tvec, inl[:ni.value].copy()


def epnp(pws, us, fu, fv, uc, vc):
    pws = _c(pws, np.float64); us = _c(us, np.float64)
    R = np.empty((3, 3)); t = np.empty(3)
    lib().orc_epnp(_p(pws), _p(us), len(pws), C.c_double(fu), C.c_double(fv), C.c_double(uc), C.c_double(vc), _p(R), _p(t))
    return R, t


def svd(A):
    A = _c(A, np.float64); m, n = A.shape; k = min(m, n)
    w = np.empty(k); u = np.empty((m, k)); vt = np.empty((k, n))
    lib().orc_svd(_p(A), m, n, _p(w), _p(u), _p(vt))
    return u, w, vt


# ------------------------------------------------------------------ stereo VO
class StereoVO:
    def __init__(self, params: VoParams, K_left, K_right, R_right, t_right, max_kpts=20000):
        self._keep = [_c(x, np.float64) for x in (K_left, K_right, R_right, t_right)]
        self.cap = max_kpts
        self._extended = bool(params.SURF_EXTENDED)
        self.dim = 128 if params.SURF_EXTENDED else 64
        self.h = lib().orc_stereo_create(C.byref(params), *[_p(x) for x in self._keep], max_kpts)

    def use_sift(self, on=True):
        """FEATURE_DETECTOR = "SIFT": the loop's detect_features / match_features take their SIFT branches (VOU:107-112, 525-529)."""
        lib().orc_stereo_use_sift.argtypes = [C.c_void_p, C.c_int]
        lib().orc_stereo_use_sift(self.h, int(bool(on)))
        self.dim = 128 if on else (128 if self._extended else 64)
        self._desc_dtype = None

    def use_detector(self, name: str, orb_pattern=None):
        """The reference's global FEATURE_DETECTOR for this loop: "SURF", "SIFT", "AKAZE" (61-byte rows, Hamming matcher) or "ORB" (32-byte
        rows, Hamming; orb_pattern = the sampling table, 256 x (x0, y0, x1, y1))."""
        det = ("SURF", "SIFT", "AKAZE", "ORB").index(name)
        pat = None
        if det == 3:
            pat = _c(np.asarray(orb_pattern).reshape(-1), np.int32)
            assert pat.size == 1024
        lib().orc_stereo_use_detector.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib().orc_stereo_use_detector(self.h, det, _p(pat) if pat is not None else None)
        self._desc_dtype = {0: np.dtype(("f4", 128 if self._extended else 64)), 1: np.dtype(("f4", 128)), 2: np.dtype(("u1", 61)), 3: np.dtype(("u1", 32))}[det]
        self.dim = self._desc_dtype.shape[0]

    def step(self, left, right, dt=0.05) -> StereoResult:
        left = _c(left, np.uint8); right = _c(right, np.uint8)
        h, w = left.shape
        r = StereoResult()
        lib().orc_stereo_step(self.h, _p(left), _p(right), w, h, w, dt, C.byref(r))
        return r

    def get(self, what: str):
        dd = getattr(self, "_desc_dtype", None) or np.dtype(("f4", self.dim))
        spec = {"kps_left": KP_DTYPE, "kps_right": KP_DTYPE, "desc_left": dd,
                "desc_right": dd, "matches_stereo": DM_DTYPE, "matches_tri": DM_DTYPE,
                "points4d": np.dtype(("f4", 4)), "good_pts": np.dtype(("f8", 3)), "good_idx": np.dtype("i4"),
                "inliers": np.dtype("i4")}[what]
        buf = np.zeros(self.cap * 4, spec)
        n = lib().orc_stereo_get(self.h, what.encode(), _p(buf), buf.nbytes)
        if n < 0:
            raise RuntimeError("buffer too small")
        out = buf[:n].copy()
        if what == "points4d":  # stored 4 x T
            raw = np.frombuffer(buf.tobytes(), np.float32)[:4 * n]
            out = raw.reshape(4, n).copy()
        return out

    def close(self):
        if self.h:
            lib().orc_stereo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ mono path
class MonoResult(C.Structure):
    _fields_ = [("published", C.c_int), ("valid", C.c_int), ("initialized", C.c_int), ("used_essential", C.c_int),
                ("success", C.c_int), ("n_kps", C.c_int), ("n_matches", C.c_int), ("n_inliers", C.c_int),
                ("n_good3d", C.c_int), ("n_front", C.c_int), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("SF", C.c_double), ("velocity", C.c_double * 3)]


def mono_params(min_hessian=50, method=4) -> VoParams:
    """mono_VO_parameters.yaml:13-49 (method 4 = LMEDS as shipped, 8 = RANSAC for BASELINE config 4)."""
    p = VoParams()
    p.DISTANCE = 10
    p.LOWE_RATIO_THRESHOLD = 0.7
    p.ESSENTIAL_OUTLIER_METHOD = method; p.ESSENTIAL_MAX_ITERS = 2000; p.ESSENTIAL_CONFIDENCE = 0.99; p.ESSENTIAL_THRESHOLD = 0.1
    p.HOMOGRAPHY_OUTLIER_METHOD = method; p.HOMOGRAPHY_MAX_ITERS = 2000; p.HOMOGRAPHY_CONFIDENCE = 0.99
    p.HOMOGRAPHY_THRESHOLD = 0.1; p.HOMOGRAPHY_DISTANCE = 50.0
    p.VPF_THRESHOLD = 0.4; p.REPROJECTION_TOLERANCE = 0.1
    p.MIN_NUM_FEATURES = 20; p.MIN_NUM_INLIERS = 10; p.MIN_NUM_3DPOINTS = 5
    p.SURF_MIN_HESSIAN = int(min_hessian); p.SURF_OCTAVES_NUMBER = 4; p.SURF_OCTAVES_LAYERS = 3
    p.SURF_EXTENDED = 0; p.SURF_UPRIGHT = 1
    return p


def solve_poly(coeffs):
    c = _c(coeffs, np.float64); n = len(c) - 1
    re = np.empty(n); im = np.empty(n)
    lib().orc_solve_poly(_p(c), n, _p(re), _p(im))
    return re + 1j * im


def jacobi_eigen(A):
    A = _c(A, np.float64).copy(); n = A.shape[0]
    w = np.empty(n); v = np.empty((n, n))
    lib().orc_jacobi_eigen(_p(A), n, _p(w), _p(v))
    return w, v


def five_point(q1, q2):
    q1 = _c(q1, np.float64); q2 = _c(q2, np.float64)
    out = np.empty((10, 9))
    n = lib().orc_five_point(_p(q1), _p(q2), _p(out))
    return out[:n].reshape(n, 3, 3).copy()


def find_essential_mat(p1, p2, K, method=8, prob=0.99, thr=0.1, max_iters=2000):
    p1 = _c(p1, np.float32); p2 = _c(p2, np.float32); K = _c(K, np.float64)
    E = np.zeros((3, 3)); mask = np.zeros(max(len(p1), 1), np.uint8)
    ok = lib().orc_find_essential_mat(_p(p1), _p(p2), len(p1), _p(K), method, C.c_double(prob), C.c_double(thr), max_iters, _p(E), _p(mask))
    return bool(ok), E, mask[:len(p1)].copy()


def recover_pose(E, p1, p2, K, mask):
    E = _c(E, np.float64); p1 = _c(p1, np.float32); p2 = _c(p2, np.float32); K = _c(K, np.float64)
    m = _c(mask, np.uint8).copy(); R = np.empty((3, 3)); t = np.empty(3)
    g = lib().orc_recover_pose(_p(E), _p(p1), _p(p2), len(p1), _p(K), _p(R), _p(t), _p(m))
    return g, R, t, m


def find_homography(p1, p2, method=8, thr=0.1, max_iters=2000, conf=0.99):
    p1 = _c(p1, np.float32); p2 = _c(p2, np.float32)
    H = np.zeros((3, 3)); mask = np.zeros(max(len(p1), 1), np.uint8)
    ok = lib().orc_find_homography(_p(p1), _p(p2), len(p1), method, C.c_double(thr), max_iters, C.c_double(conf), _p(H), _p(mask))
    return bool(ok), H, mask[:len(p1)].copy()


def decompose_homography(H, K):
    H = _c(H, np.float64); K = _c(K, np.float64)
    Rs = np.empty((4, 3, 3)); ts = np.empty((4, 3)); ns = np.empty((4, 3))
    n = lib().orc_decompose_homography_mat(_p(H), _p(K), _p(Rs), _p(ts), _p(ns))
    return Rs[:n].copy(), ts[:n].copy(), ns[:n].copy()


def recover_pose_homography(H, p1, p2, K, dist=50.0):
    H = _c(H, np.float64); p1 = _c(p1, np.float32); p2 = _c(p2, np.float32); K = _c(K, np.float64)
    R = np.full((3, 3), np.nan); t = np.full(3, np.nan)
    g = lib().orc_recover_pose_homography(_p(H), _p(p1), _p(p2), len(p1), _p(K), C.c_double(dist), _p(R), _p(t))
    return g, R, t


def estimate_relative_pose(params: VoParams, use_essential: bool, p1, p2, K, R0=None, t0=None):
    p1 = _c(p1, np.float32); p2 = _c(p2, np.float32); K = _c(K, np.float64)
    n = len(p1)
    R = np.eye(3) if R0 is None else _c(R0, np.float64).copy()
    t = np.zeros(3) if t0 is None else _c(t0, np.float64).copy()
    in1 = np.empty((max(n, 1), 2), np.float32); in2 = np.empty((max(n, 1), 2), np.float32)
    ue = C.c_int(int(use_essential)); nin = C.c_int(0); mask = np.zeros(max(n, 1), np.uint8)
    ok = lib().orc_estimate_relative_pose(C.byref(params), C.byref(ue), _p(p1), _p(p2), n, _p(K), _p(R), _p(t), _p(in1), _p(in2),
                                          C.byref(nin), _p(mask))
    return bool(ok), bool(ue.value), R, t, in1[:nin.value].copy(), in2[:nin.value].copy(), mask[:n].copy()


class MonoVO:
    def __init__(self, params: VoParams, K, max_kpts=20000):
        self._K = _c(K, np.float64)
        self.cap = max_kpts
        lib().orc_mono_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_mono_create(C.byref(params), _p(self._K), max_kpts))

    def use_sift(self, on=True):
        """FEATURE_DETECTOR = "SIFT" for the mono loop (VOU:107-112, 525-529)."""
        lib().orc_mono_use_sift.argtypes = [C.c_void_p, C.c_int]
        lib().orc_mono_use_sift(self.h, int(bool(on)))

    def use_detector(self, name: str, orb_pattern=None):
        """FEATURE_DETECTOR for the mono loop: "SURF", "SIFT", "AKAZE" or "ORB" (orb_pattern = the sampling table, 256 x (x0, y0, x1, y1))."""
        det = ("SURF", "SIFT", "AKAZE", "ORB").index(name)
        pat = _c(np.asarray(orb_pattern).reshape(-1), np.int32) if det == 3 else None
        lib().orc_mono_use_detector.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib().orc_mono_use_detector(self.h, det, _p(pat) if pat is not None else None)

    def step(self, img, rng=1.0, dt=0.05) -> MonoResult:
        img = _c(img, np.uint8)
        h, w = img.shape
        r = MonoResult()
        lib().orc_mono_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        lib().orc_mono_step(self.h, _p(img), w, h, w, float(rng), float(dt), C.byref(r))
        return r

    def get(self, what):
        spec = {"kps": KP_DTYPE, "matches": DM_DTYPE, "mask": np.dtype("u1"), "good_pts": np.dtype(("f8", 3))}[what]
        buf = np.zeros(self.cap, spec)
        lib().orc_mono_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        n = lib().orc_mono_get(self.h, what.encode(), _p(buf), buf.nbytes)
        return buf[:max(n, 0)].copy()

    def close(self):
        if self.h:
            lib().orc_mono_destroy.argtypes = [C.c_void_p]
            lib().orc_mono_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- get_image preprocessing
def rgb2gray(rgb):
    rgb = _c(rgb, np.uint8); h, w, _ = rgb.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_rgb2gray_u8(_p(rgb), w, h, w * 3, _p(out))
    return out


def resize_area_c3(rgb, dw, dh):
    rgb = _c(rgb, np.uint8); h, w, _ = rgb.shape
    out = np.empty((dh, dw, 3), np.uint8)
    if lib().orc_resize_area_u8c3(_p(rgb), w, h, w * 3, _p(out), dw, dh) != 0:
        raise ValueError("resize_area_c3: enlarging is outside the restatement")
    return out


def undistort(gray, K, dist4, newK):
    gray = _c(gray, np.uint8); h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_undistort_u8(_p(gray), w, h, _p(_c(K, np.float64)), _p(_c(dist4, np.float64)), _p(_c(newK, np.float64)), _p(out))
    return out


def clahe(gray, clip_limit):
    gray = _c(gray, np.uint8); h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_clahe_u8(_p(gray), w, h, C.c_double(clip_limit), _p(out))
    return out


def get_image(rgb, desired_width, K, dist4, newK, clahe_on=True, clip_limit=3):
    rgb = _c(rgb, np.uint8); h, w, _ = rgb.shape
    dh = int(h / (w / desired_width))
    out = np.empty((dh, desired_width), np.uint8)
    ow, oh = C.c_int(0), C.c_int(0)
    rc = lib().orc_get_image(_p(rgb), w, h, w * 3, desired_width, _p(_c(K, np.float64)), _p(_c(dist4, np.float64)),
                             _p(_c(newK, np.float64)), int(bool(clahe_on)), int(clip_limit), _p(out), C.byref(ow), C.byref(oh))
    if rc != 0:
        raise ValueError("get_image: enlarging is outside the restatement")
    assert (ow.value, oh.value) == (desired_width, dh)
    return out


def resize_camera_matrix(original_width, original_height, desired_width, K, dist4):
    """-> (K scaled, newK, desired_height): reference VO_utility.cpp:658-675."""
    Ks = np.array(K, np.float64).reshape(3, 3).copy()
    newK = np.zeros((3, 3), np.float64)
    dh = C.c_int(0)
    rc = lib().orc_resize_camera_matrix(int(original_width), int(original_height), int(desired_width), _p(Ks), _p(_c(dist4, np.float64)),
                                        _p(newK), C.byref(dh))
    if rc != 0:
        raise ValueError("resize_camera_matrix: bad sizes")
    return Ks, newK, dh.value


# ---------------------------------------------------------------- compressed-image ingest
def jpeg_decode(data: bytes) -> np.ndarray:
    """cv::imdecode(IMREAD_UNCHANGED) of a baseline JPEG: H x W x 3 BGR, or H x W grey."""
    buf = np.frombuffer(data, np.uint8)
    w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
    lib().orc_jpeg_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib().orc_jpeg_decode(_p(buf), len(buf), None, 0, C.byref(w), C.byref(h), C.byref(ch))
    if rc != 0:
        raise ValueError(f"jpeg_decode: error {rc}")
    out = np.empty((h.value, w.value, ch.value), np.uint8)
    rc = lib().orc_jpeg_decode(_p(buf), len(buf), _p(out), out.nbytes, C.byref(w), C.byref(h), C.byref(ch))
    if rc != 0:
        raise ValueError(f"jpeg_decode: error {rc}")
    return out[..., 0] if ch.value == 1 else out


def sift_detect(img: np.ndarray, nfeatures=10000, n_octave_layers=3, contrast_threshold=0.03, edge_threshold=10.0, sigma=1.6, cap=1 << 16,
                descriptors=True):
    """SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute (VO_utility.cpp:107-112), restated: (keypoints, n x 128 descriptors)."""
    img = _c(img, np.uint8)
    h, w = img.shape
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 128), np.float32) if descriptors else None
    f = lib().orc_sift_detect_and_compute
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    n = f(_p(img), w, h, w, int(nfeatures), int(n_octave_layers), float(contrast_threshold), float(edge_threshold), float(sigma),
          _p(kps), _p(desc) if descriptors else None, cap)
    if n < 0:
        raise ValueError(f"sift_detect: capacity {cap} too small for {-n} keypoints")
    return kps[:n].copy(), (desc[:n].copy() if descriptors else None)


def sift_gauss_layer(img: np.ndarray, octave: int, layer: int, n_octave_layers=3, sigma=1.6) -> np.ndarray:
    img = _c(img, np.uint8)
    h, w = img.shape
    ow, oh = C.c_int(0), C.c_int(0)
    f = lib().orc_sift_gauss_layer
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    out = np.zeros((2 * h + 2) * (2 * w + 2), np.float32)
    n = f(_p(img), w, h, w, int(n_octave_layers), float(sigma), int(octave), int(layer), _p(out), C.byref(ow), C.byref(oh))
    if n == 0:
        raise ValueError("sift_gauss_layer: no such layer")
    return out[:n].reshape(oh.value, ow.value).copy()


def png_decode(data: bytes) -> np.ndarray:
    """cv::imdecode(IMREAD_UNCHANGED) of an 8-bit PNG: H x W grey, H x W x 3 BGR (RGB or palette) or H x W x 4 BGRA."""
    buf = np.frombuffer(data, np.uint8)
    w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
    lib().orc_png_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib().orc_png_decode(_p(buf), len(buf), None, 0, C.byref(w), C.byref(h), C.byref(ch))
    if rc != 0:
        raise ValueError(f"png_decode: error {rc}")
    out = np.empty((h.value, w.value, ch.value), np.uint8)
    rc = lib().orc_png_decode(_p(buf), len(buf), _p(out), out.nbytes, C.byref(w), C.byref(h), C.byref(ch))
    if rc != 0:
        raise ValueError(f"png_decode: error {rc}")
    return out[..., 0] if ch.value == 1 else out


def bayer_bggr2bgr(bayer: np.ndarray) -> np.ndarray:
    bayer = _c(bayer, np.uint8); h, w = bayer.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_bayer_bggr2bgr(_p(bayer), w, h, w, _p(out))
    return out


# ---------------------------------------------------------------------------------------------- AKAZE (o_akaze.c)
AKAZE_DESC_BYTES = 61


def akaze_detect(img: np.ndarray, cap=1 << 16, descriptors=True):
    """AKAZE::create()->detectAndCompute (VO_utility.cpp:93-98), restated: (keypoints, n x 61 uint8 M-LDB descriptors)."""
    img = _c(img, np.uint8)
    h, w = img.shape
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, AKAZE_DESC_BYTES), np.uint8) if descriptors else None
    f = lib().orc_akaze_detect_and_compute
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    n = f(_p(img), w, h, w, _p(kps), _p(desc) if descriptors else None, cap)
    if n < 0:
        raise ValueError(f"akaze_detect: capacity {cap} too small for {-n} keypoints")
    return kps[:n].copy(), (desc[:n].copy() if descriptors else None)


def akaze_fed_tau(T: float, tau_max: float = 0.25) -> np.ndarray:
    tau = np.zeros(256, np.float32)
    f = lib().orc_akaze_fed_tau
    f.argtypes = [C.c_float, C.c_float, C.c_void_p]
    n = f(C.c_float(T), C.c_float(tau_max), _p(tau))
    return tau[:n].copy()


def akaze_levels(w: int, h: int):
    """-> (array [n][6]: w, h, octave, sigma_size, border, FED steps into the level; esigma [n])"""
    out = np.zeros((16, 6), np.int32); es = np.zeros(16, np.float32)
    f = lib().orc_akaze_levels
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = f(int(w), int(h), _p(out), _p(es))
    return out[:n].copy(), es[:n].copy()


def akaze_plane(img: np.ndarray, level: int, what: int):
    """what: 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly (multiscale derivatives), 4 Ldet of evolution level `level` -> (plane, kcontrast)"""
    img = _c(img, np.uint8)
    h, w = img.shape
    out = np.zeros(h * w, np.float32)
    ow, oh, kc = C.c_int(0), C.c_int(0), C.c_float(0)
    f = lib().orc_akaze_plane
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = f(_p(img), w, h, w, int(level), int(what), _p(out), C.byref(ow), C.byref(oh), C.byref(kc))
    if n == 0:
        raise ValueError("akaze_plane: no such level")
    return out[:n].reshape(oh.value, ow.value).copy(), kc.value


def akaze_scharr(src: np.ndarray, xorder: bool) -> np.ndarray:
    src = _c(src, np.float32); out = np.zeros_like(src)
    f = lib().orc_akaze_scharr
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f(_p(src), src.shape[1], src.shape[0], int(bool(xorder)), _p(out))
    return out


def akaze_pm_g2(lx: np.ndarray, ly: np.ndarray, k: float) -> np.ndarray:
    lx, ly = _c(lx, np.float32), _c(ly, np.float32); out = np.zeros_like(lx)
    f = lib().orc_akaze_pm_g2
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    f(_p(lx), _p(ly), lx.size, C.c_float(k), _p(out))
    return out


def akaze_nld_step(lt: np.ndarray, lf: np.ndarray, step: float) -> np.ndarray:
    lt, lf = _c(lt, np.float32), _c(lf, np.float32); out = np.zeros_like(lt)
    f = lib().orc_akaze_nld_step
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    f(_p(lt), _p(lf), lt.shape[1], lt.shape[0], C.c_float(step), _p(out))
    return out


# ---------------------------------------------------------------------------------------------- ORB (o_orb.c)
ORB_DESC_BYTES = 32
# ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10), VO_utility.cpp:103
ORB_REFERENCE = dict(nfeatures=10000, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, patchSize=31, fastThreshold=10)


def orb_random_pattern(patch_size=31, npoints=512) -> np.ndarray:
    """orb.cpp makeRandomPattern: the (npoints, 2) int32 table OpenCV draws for patch sizes other than 31 (cv::RNG(0x34985739))."""
    out = np.zeros((npoints, 2), np.int32)
    f = lib().orc_orb_random_pattern
    f.argtypes = [C.c_int, C.c_void_p, C.c_int]
    f(int(patch_size), _p(out), int(npoints))
    return out


def orb_levels(w, h, nfeatures=10000, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, patchSize=31):
    """-> (border, [(w, h, features wanted)] per level, scales)"""
    out = np.zeros((nlevels, 3), np.int32); sc = np.zeros(nlevels, np.float32)
    f = lib().orc_orb_levels
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    b = f(w, h, nfeatures, C.c_float(scaleFactor), nlevels, edgeThreshold, firstLevel, patchSize, _p(out), _p(sc))
    return b, out, sc


def orb_detect(img: np.ndarray, pattern=None, cap=1 << 16, nfeatures=10000, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0,
               patchSize=31, fastThreshold=10):
    """ORB::create(...)->detectAndCompute (VO_utility.cpp:100-105), restated: (keypoints, n x 32 uint8 rBRIEF rows or None without a pattern)."""
    img = _c(img, np.uint8)
    h, w = img.shape
    kps = np.zeros(cap, KP_DTYPE)
    pat = None if pattern is None else _c(np.asarray(pattern).reshape(-1), np.int32)
    if pat is not None and pat.size != 1024:
        raise ValueError("orb_detect: the pattern is 256 x (x0, y0, x1, y1)")
    desc = np.zeros((cap, ORB_DESC_BYTES), np.uint8) if pat is not None else None
    f = lib().orc_orb_detect_and_compute
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = f(_p(img), w, h, w, nfeatures, C.c_float(scaleFactor), nlevels, edgeThreshold, firstLevel, patchSize, fastThreshold,
          _p(pat) if pat is not None else None, _p(kps), _p(desc) if desc is not None else None, cap)
    if n < 0:
        raise ValueError(f"orb_detect: capacity {cap} too small for {-n} keypoints")
    return kps[:n].copy(), (desc[:n].copy() if desc is not None else None)


def resize_linear_exact(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = _c(src, np.uint8); out = np.zeros((dh, dw), np.uint8)
    f = lib().orc_resize_linear_exact_u8
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    f(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(out), dw, dh, dw)
    return out


def fast_scores(img: np.ndarray, threshold: int) -> np.ndarray:
    img = _c(img, np.uint8); out = np.zeros_like(img)
    f = lib().orc_fast_scores
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f(_p(img), img.shape[1], img.shape[0], img.shape[1], int(threshold), _p(out))
    return out


def fast_detect(img: np.ndarray, threshold: int, cap=1 << 18) -> np.ndarray:
    img = _c(img, np.uint8); kps = np.zeros(cap, KP_DTYPE)
    f = lib().orc_fast_detect
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = f(_p(img), img.shape[1], img.shape[0], img.shape[1], int(threshold), _p(kps), cap)
    if n < 0:
        raise ValueError("fast_detect: capacity")
    return kps[:n].copy()


def retain_best(responses: np.ndarray, n_points: int) -> np.ndarray:
    """KeyPointsFilter::retainBest on a vector with these responses: the surviving old indices in their new order."""
    r = _c(responses, np.float32); perm = np.zeros(max(r.size, 1), np.int32)
    f = lib().orc_retain_best
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    m = f(_p(r), r.size, int(n_points), _p(perm))
    return perm[:m].copy()


def orb_harris(img: np.ndarray, x: int, y: int) -> float:
    img = _c(img, np.uint8)
    f = lib().orc_orb_harris
    f.restype = C.c_float; f.argtypes = [C.c_void_p, C.c_int]
    return f(C.c_void_p(img.ctypes.data + y * img.shape[1] + x), img.shape[1])


def orb_umax(half=15) -> np.ndarray:
    u = np.zeros(half + 2, np.int32)
    f = lib().orc_orb_umax
    f.argtypes = [C.c_int, C.c_void_p]
    f(half, _p(u))
    return u


def orb_ic_angle(img: np.ndarray, x: int, y: int, half=15) -> float:
    img = _c(img, np.uint8); u = orb_umax(half)
    f = lib().orc_orb_ic_angle
    f.restype = C.c_float; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return f(C.c_void_p(img.ctypes.data + y * img.shape[1] + x), img.shape[1], _p(u), half)


def orb_blur_kernel() -> np.ndarray:
    k = np.zeros(7, np.int32)
    f = lib().orc_orb_blur_kernel
    f.argtypes = [C.c_void_p]
    f(_p(k))
    return k


def orb_blur(img: np.ndarray) -> np.ndarray:
    img = _c(img, np.uint8); out = np.zeros_like(img)
    f = lib().orc_orb_blur_u8
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    f(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def orb_describe(img: np.ndarray, x: int, y: int, angle_deg: float, pattern) -> np.ndarray:
    img = _c(img, np.uint8); pat = _c(np.asarray(pattern).reshape(-1), np.int32); out = np.zeros(32, np.uint8)
    f = lib().orc_orb_describe
    f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    f(C.c_void_p(img.ctypes.data + y * img.shape[1] + x), img.shape[1], C.c_float(angle_deg), _p(pat), _p(out))
    return out


def orb_level_image(img: np.ndarray, level: int, blurred=False, scaleFactor=1.2, nlevels=8) -> np.ndarray:
    img = _c(img, np.uint8); h, w = img.shape
    out = np.zeros(h * w, np.uint8); ow, oh = C.c_int(0), C.c_int(0)
    f = lib().orc_orb_level_image
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = f(_p(img), w, h, w, C.c_float(scaleFactor), nlevels, int(level), int(bool(blurred)), _p(out), out.size, C.byref(ow), C.byref(oh))
    if rc != 0:
        raise ValueError("orb_level_image")
    return out[:ow.value * oh.value].reshape(oh.value, ow.value).copy()
