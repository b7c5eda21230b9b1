"""ergo_uvo_amd -- MI355X-native hot path of UVO (team-ergo-unipi/ergo_uvo): upright SURF-64,
brute-force L2 2-NN + ratio test, triangulation and EPnP PnP-RANSAC, behind the C ABI of
include/uvo_hip.h (hand-written HIP for gfx950 in csrc/).

This module is the Python-side mirror of the reference's `uvo_libraries` function surface
(uvo_libraries/include/uvo_libraries/VO_utility.h:96-117): same function names and argument
meaning, numpy in / numpy out, every call going through libuvo_hip.so.  Images and descriptors
may also be torch CUDA(ROCm) tensors, in which case they are used in place (no host copy).
"""
from __future__ import annotations

import collections
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["Params", "Context", "StereoResult", "MonoResult", "UvoError", "KP_DTYPE", "DM_DTYPE", "build"]

build = _lib.build

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])          # cv::KeyPoint, 28 B
DM_DTYPE = np.dtype([("queryIdx", "i4"), ("trainIdx", "i4"), ("imgIdx", "i4"), ("distance", "f4")])  # cv::DMatch

MEM_HOST, MEM_DEVICE = 0, 1
_STATUS = {1: "UVO_INVALID_ARG", 2: "UVO_TOO_FEW_POINTS", 3: "UVO_CAPACITY", 4: "UVO_HIP_ERROR", 5: "UVO_NO_DEVICE"}


class UvoError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"{_STATUS.get(status, status)}: {msg}")
        self.status = status


class Params(C.Structure):
    """uvo_params: the globals of VO_utility.h:25-89 read by the hot path."""
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

    @classmethod
    def stereo(cls, **kw) -> "Params":
        p = cls()
        _lib.lib().uvo_params_default_stereo(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    @classmethod
    def mono(cls, **kw) -> "Params":
        p = cls()
        _lib.lib().uvo_params_default_mono(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class StereoResult(C.Structure):
    _fields_ = [("valid", C.c_int), ("initialized", C.c_int), ("n_left", C.c_int), ("n_right", C.c_int),
                ("n_stereo_matches", C.c_int), ("n_tri_matches", C.c_int), ("n_good3d", C.c_int),
                ("n_inliers", C.c_int), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3),
                ("t_prev_curr", C.c_double * 3), ("velocity", C.c_double * 3)]


class MonoResult(C.Structure):
    _fields_ = [("published", C.c_int), ("valid", C.c_int), ("initialized", C.c_int), ("used_essential", C.c_int),
                ("success", C.c_int), ("n_kps", C.c_int), ("n_matches", C.c_int), ("n_inliers", C.c_int),
                ("n_good3d", C.c_int), ("n_front", C.c_int), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("SF", C.c_double), ("velocity", C.c_double * 3)]


def _is_device(a) -> bool:
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _ptr_mem(a, dtype):
    """(pointer, mem, keepalive) of a numpy array or a torch CUDA tensor."""
    if _is_device(a):
        if not a.is_contiguous():
            raise ValueError("device tensors must be contiguous")
        return C.c_void_p(a.data_ptr()), MEM_DEVICE, a
    arr = np.ascontiguousarray(a, dtype=dtype)
    return arr.ctypes.data_as(C.c_void_p), MEM_HOST, arr


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


TRACE_DTYPE = np.dtype([("pair", "i8"), ("lane", "i4"), ("b_used", "i4"), ("dev_ms", "f4", 8), ("host_ms", "f8", 6)], align=True)   # uvo_trace_row


class Context:
    """One uvo_ctx: one GPU, one HIP stream, all device workspaces."""

    def __init__(self, params: Params, device: int = 0, max_w: int = 1920, max_h: int = 1080, max_kpts: int = 8192):
        self._lib = _lib.lib()
        self.params = params
        self.max_kpts = max_kpts
        self.max_w, self.max_h = max_w, max_h
        h = C.c_void_p()
        st = self._lib.uvo_ctx_create(C.byref(params), device, max_w, max_h, max_kpts, C.byref(h))
        if st != 0:
            raise UvoError(st, "uvo_ctx_create failed (no usable HIP device?)" if st == 5 else "uvo_ctx_create failed")
        self._h = h
        self._inflight = collections.deque()       # tensors handed to submit: kept alive until the matching collect
        self._producer = None
        self.set_producer_stream("torch")           # device tensors are ordered after the torch stream that is current at each call

    # ------------------------------------------------------------------ plumbing
    def _check(self, st):
        if st != 0:
            raise UvoError(st, (self._lib.uvo_last_error(self._h) or b"").decode())

    def _release_collected(self, pending_before: int):
        """Drop the keep-alive entry of a pair / frame only if the C call really dequeued it (a collect refused with "nothing
        submitted" or "wrong kind" leaves the queue alone, and the images of the oldest entry may still be read in place)."""
        if self._lib.uvo_ctx_pending(self._h) < pending_before and self._inflight:
            self._inflight.popleft()

    def _drained(self):
        """reset / set_rig / set_depth drain the C pipeline: nothing reads the submitted tensors any more."""
        if self._lib.uvo_ctx_pending(self._h) == 0:
            self._inflight.clear()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.uvo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return self._lib.uvo_ctx_stream(self._h)

    @property
    def warning(self) -> str:
        """Advice about the process environment noticed by the library (e.g. GPU_MAX_HW_QUEUES too small for the pipeline depth)."""
        return (self._lib.uvo_ctx_warning(self._h) or b"").decode()

    def host_policy(self) -> dict:
        """uvo_ctx_host_policy: how the context's host side waits and who drives the PnP stage of pipelined pairs, e.g.
        {"wait": "poll", "stage_b": "device", "cpu_budget": 2.0, "depth": 6}."""
        out = {}
        for kv in (self._lib.uvo_ctx_host_policy(self._h) or b"").decode().split():
            k, _, v = kv.partition("=")
            out[k] = float(v) if k == "cpu_budget" else (int(v) if k == "depth" else v)
        return out

    def set_producer_stream(self, stream):
        """Declare the stream that produces device inputs: a raw hipStream_t (int, 0 = the default stream), a torch.cuda.Stream,
        "torch" for torch's current stream at each call, or None to switch the ordering off (inputs must then be complete
        before each call).  Tensors handed to stereo_submit / mono_submit are kept alive until the matching collect."""
        self._producer = stream
        if stream is None:
            self._check(self._lib.uvo_ctx_set_producer_stream(self._h, None, 0))
        elif stream != "torch":
            self._check(self._lib.uvo_ctx_set_producer_stream(self._h, C.c_void_p(int(getattr(stream, "cuda_stream", stream)) or None), 1))

    def _order_after_producer(self, *arrays):
        if self._producer == "torch" and any(_is_device(a) for a in arrays):
            import torch
            self._check(self._lib.uvo_ctx_set_producer_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream or None), 1))

    def set_feature_detector(self, name: str):
        """The reference's global FEATURE_DETECTOR: "SURF" (default) or "SIFT" for the fused steps and detect_features; "AKAZE" for
        detect_features / match_features alone (the fused steps refuse it)."""
        if name in ("AKAZE", "ORB"):
            self._feature_akaze, self._feature_orb, self._feature_sift = name == "AKAZE", name == "ORB", False
            return
        self._check(self._lib.uvo_ctx_set_feature_detector(self._h, name.encode()))
        self._feature_sift = name == "SIFT"
        self._feature_akaze = self._feature_orb = False

    def set_params(self, params: Params):
        self.params = params
        self._check(self._lib.uvo_ctx_set_params(self._h, C.byref(params)))

    # ------------------------------------------------------------------ uvo_libraries mirror
    def detect_features(self, img):
        """detect_features(img, keypoints, descriptors): the SURF branch (VO_utility.cpp:114-119), or the SIFT branch (VO_utility.cpp:107-112)
        when set_feature_detector("SIFT") was called -- the reference switches on its global FEATURE_DETECTOR."""
        if getattr(self, "_feature_akaze", False):
            return self.akaze_detect(img)                  # VO_utility.cpp:93-98
        if getattr(self, "_feature_orb", False):
            return self.orb_detect(img)                    # VO_utility.cpp:100-105
        if getattr(self, "_feature_sift", False):
            return self.sift_detect(img)
        return self.surf_detect(img)

    def akaze_detect(self, img, cap=None):
        """detect_features, FEATURE_DETECTOR == "AKAZE" (VO_utility.cpp:93-98): AKAZE::create()->detectAndCompute.
        Returns (keypoints, n x 61 uint8 M-LDB descriptors) -- rows for match_features_hamming."""
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        cap = int(cap if cap is not None else self.max_kpts)
        n = C.c_int(0)
        kps = np.empty(cap, KP_DTYPE)
        desc = np.empty((cap, 61), np.uint8)
        self._order_after_producer(img)
        self._check(self._lib.uvo_akaze_detect(self._h, p, w, h, w, mem, _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value], desc[:n.value]

    def akaze_plane(self, level: int, what: int):
        """Test hook: plane `what` (0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet) of evolution level `level` of the last akaze_detect."""
        w, h = C.c_int(0), C.c_int(0)
        out = np.empty(self.max_w * self.max_h, np.float32)
        self._check(self._lib.uvo_akaze_plane(self._h, int(level), int(what), _p(out), out.size, C.byref(w), C.byref(h)))
        return out[:w.value * h.value].reshape(h.value, w.value).copy()

    def orb_configure(self, nfeatures=10000, scaleFactor=1.2, nlevels=8, edgeThreshold=31, patchSize=31, fastThreshold=10):
        """The arguments of ORB::create the reference passes (VO_utility.cpp:103) are the defaults; firstLevel 0, WTA_K 2, HARRIS_SCORE are fixed."""
        f = self._lib.uvo_orb_configure
        f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        self._check(f(self._h, int(nfeatures), float(scaleFactor), int(nlevels), int(edgeThreshold), int(patchSize), int(fastThreshold)))

    def orb_set_pattern(self, pattern):
        """The rBRIEF sampling table in OpenCV's bit_pattern_31_ layout (256 x (x0, y0, x1, y1)); None forgets it."""
        if pattern is None:
            self._check(self._lib.uvo_orb_set_pattern(self._h, None))
            return
        pat = np.ascontiguousarray(np.asarray(pattern).reshape(-1), np.int32)
        if pat.size != 1024:
            raise ValueError("orb_set_pattern: 256 x (x0, y0, x1, y1)")
        self._check(self._lib.uvo_orb_set_pattern(self._h, _p(pat)))

    def orb_detect(self, img, cap=None, descriptors=True):
        """detect_features, FEATURE_DETECTOR == "ORB" (VO_utility.cpp:100-105): ORB::create(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, 10)
        ->detectAndCompute.  Returns (keypoints, n x 32 uint8 rBRIEF rows) -- rows for match_features_hamming; descriptors need
        orb_set_pattern (descriptors=False: keypoints only)."""
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        cap = int(cap if cap is not None else self.max_kpts)
        n = C.c_int(0)
        kps = np.empty(cap, KP_DTYPE)
        desc = np.empty((cap, 32), np.uint8) if descriptors else None
        self._order_after_producer(img)
        self._check(self._lib.uvo_orb_detect(self._h, p, w, h, w, mem, _p(kps), _p(desc) if descriptors else None, cap, C.byref(n)))
        return kps[:n.value], (desc[:n.value] if descriptors else None)

    def orb_plane(self, level: int, what: int):
        """Test hook: level `level` of the last orb_detect: what = 0 the resized image, 1 its blurred copy, 2 the FAST score map."""
        w, h = C.c_int(0), C.c_int(0)
        out = np.empty(self.max_w * self.max_h, np.uint8)
        self._check(self._lib.uvo_orb_plane(self._h, int(level), int(what), _p(out), out.size, C.byref(w), C.byref(h)))
        return out[:w.value * h.value].reshape(h.value, w.value).copy()

    def surf_detect(self, img):
        """uvo_surf_detect: the SURF operator itself (64- or, with SURF_EXTENDED, 128-float rows), whatever set_feature_detector says."""
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        n = C.c_int(0)
        kps = np.zeros(self.max_kpts, KP_DTYPE)
        desc = np.zeros((self.max_kpts, 128 if self.params.SURF_EXTENDED else 64), np.float32)
        self._order_after_producer(img)
        self._check(self._lib.uvo_surf_detect(self._h, p, w, h, w, mem, _p(kps), _p(desc), self.max_kpts, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def sift_detect(self, img, nfeatures=10000, n_octave_layers=3, contrast_threshold=0.03, edge_threshold=10.0, sigma=1.6, cap=None):
        """detect_features, FEATURE_DETECTOR == "SIFT" (VO_utility.cpp:107-112): SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute.
        Returns (keypoints, n x 128 float descriptors)."""
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        cap = int(cap if cap is not None else max(self.max_kpts, nfeatures if nfeatures > 0 else 0))
        n = C.c_int(0)
        kps = np.empty(cap, KP_DTYPE)                     # fresh buffers, returned as views of their first n rows (no second 5 MB copy)
        desc = np.empty((cap, 128), np.float32)
        self._order_after_producer(img)
        self._check(self._lib.uvo_sift_detect(self._h, p, w, h, w, mem, int(nfeatures), int(n_octave_layers), float(contrast_threshold),
                                              float(edge_threshold), float(sigma), _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value], desc[:n.value]

    def sift_layer(self, octave: int, layer: int, dog: bool = False):
        """Test hook: a Gaussian / DoG layer of the last sift_detect (octave 0 = the doubled image)."""
        w, h = C.c_int(0), C.c_int(0)
        self._check(self._lib.uvo_sift_layer(self._h, int(octave), int(layer), int(bool(dog)), None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.float32)
        self._check(self._lib.uvo_sift_layer(self._h, int(octave), int(layer), int(bool(dog)), _p(out), out.size, C.byref(w), C.byref(h)))
        return out

    def integral(self, img):
        h, w = img.shape
        p, mem, keep = _ptr_mem(img, np.uint8)
        out = np.empty((h + 1, w + 1), np.int32)
        self._check(self._lib.uvo_integral(self._h, p, w, h, w, mem, _p(out)))
        return out

    def hessian_layer(self, shape, octave: int, layer: int):
        h, w = shape
        step = 1 << octave
        det = np.empty((h // step, w // step), np.float32)
        tr = np.empty_like(det)
        self._check(self._lib.uvo_hessian_layer(self._h, octave, layer, _p(det), _p(tr)))
        return det, tr

    def match_features(self, descriptors1, descriptors2, ratio=None, matches=None, dim=None):
        """match_features (VO_utility.cpp:515-543); appends to `matches` like the reference.  dim = 128: the "SIFT" arm of the
        L2 branch (VO_utility.cpp:525-529), rows of 128 floats whatever SURF_EXTENDED says."""
        ratio = float(self.params.LOWE_RATIO_THRESHOLD if ratio is None else ratio)
        self._check_desc_width(descriptors1, descriptors2, dim=dim)
        n1, n2 = int(descriptors1.shape[0]), int(descriptors2.shape[0])
        p1, m1, k1 = _ptr_mem(descriptors1, np.float32)
        p2, m2, k2 = _ptr_mem(descriptors2, np.float32)
        if m1 != m2:
            raise ValueError("descriptors1 and descriptors2 must live in the same memory space")
        prev = 0 if matches is None else len(matches)
        out = np.zeros(prev + max(n1, 1), DM_DTYPE)
        if prev:
            out[:prev] = matches
        m = C.c_int(prev)
        if dim is None:
            self._check(self._lib.uvo_match_knn2_ratio(self._h, p1, n1, p2, n2, m1, C.c_float(ratio), _p(out), len(out), C.byref(m)))
        else:
            self._check(self._lib.uvo_match_knn2_ratio_dim(self._h, p1, n1, p2, n2, int(dim), m1, C.c_float(ratio), _p(out), len(out), C.byref(m)))
        return out[:m.value].copy()

    def match_features_hamming(self, descriptors1, descriptors2, ratio=None, matches=None):
        """The AKAZE / ORB branch of match_features (VO_utility.cpp:520-524): uint8 rows of up to 64 bytes, Hamming distance."""
        ratio = float(self.params.LOWE_RATIO_THRESHOLD if ratio is None else ratio)
        d1, d2 = _np(descriptors1, np.uint8), _np(descriptors2, np.uint8)
        if d1.ndim != 2 or d2.ndim != 2 or d1.shape[1] != d2.shape[1]:
            raise ValueError("binary descriptors: two uint8 matrices with the same number of columns")
        prev = 0 if matches is None else len(matches)
        out = np.zeros(prev + max(len(d1), 1), DM_DTYPE)
        if prev:
            out[:prev] = matches
        m = C.c_int(prev)
        self._check(self._lib.uvo_match_knn2_ratio_hamming(self._h, _p(d1), len(d1), _p(d2), len(d2), d1.shape[1], 0, C.c_float(ratio), _p(out), len(out), C.byref(m)))
        return out[:m.value].copy()

    def knn_match_hamming(self, descriptors1, descriptors2):
        d1, d2 = _np(descriptors1, np.uint8), _np(descriptors2, np.uint8)
        idx = np.empty((len(d1), 2), np.int32)
        dist = np.empty((len(d1), 2), np.float32)
        self._check(self._lib.uvo_match_knn2_hamming(self._h, _p(d1), len(d1), _p(d2), len(d2), d1.shape[1], 0, _p(idx), _p(dist)))
        return idx, dist

    def _check_desc_width(self, *descs, dim=None):
        dim = dim or (128 if self.params.SURF_EXTENDED else 64)
        for d in descs:
            if d.ndim != 2 or int(d.shape[1]) != dim:
                raise ValueError("descriptor rows must have %d elements: without `dim` the standalone matchers take this context's SURF rows "
                                 "(SURF_EXTENDED = %d), whatever detector the fused steps use; pass dim=128 for SIFT rows" % (dim, int(self.params.SURF_EXTENDED)))

    def knn_match(self, descriptors1, descriptors2, dim=None):
        self._check_desc_width(descriptors1, descriptors2, dim=dim)
        n1, n2 = int(descriptors1.shape[0]), int(descriptors2.shape[0])
        p1, m1, k1 = _ptr_mem(descriptors1, np.float32)
        p2, m2, k2 = _ptr_mem(descriptors2, np.float32)
        idx = np.empty((n1, 2), np.int32)
        dist = np.empty((n1, 2), np.float32)
        if dim is None:
            self._check(self._lib.uvo_match_knn2(self._h, p1, n1, p2, n2, m1, _p(idx), _p(dist)))
        else:
            self._check(self._lib.uvo_match_knn2_dim(self._h, p1, n1, p2, n2, int(dim), m1, _p(idx), _p(dist)))
        return idx, dist

    def triangulatePoints(self, P1, P2, pts1, pts2):
        P1, P2 = _np(P1, np.float64), _np(P2, np.float64)
        x1, x2 = _np(pts1, np.float32), _np(pts2, np.float32)
        n = len(x1)
        out = np.empty((4, n), np.float32)
        self._check(self._lib.uvo_triangulate_points(self._h, _p(P1), _p(P2), _p(x1), _p(x2), n, _p(out)))
        return out

    def extract_3Dpoints(self, k1, k2, R1, t1, R2, t2, K1, K2, points4D):
        k1, k2 = _np(k1, np.float32), _np(k2, np.float32)
        n = len(k1)
        p4 = _np(points4D, np.float32)
        a = [_np(x, np.float64) for x in (R1, t1, R2, t2, K1, K2)]
        pts = np.empty((max(n, 1), 3))
        idx = np.empty(max(n, 1), np.int32)
        g = C.c_int(0)
        self._check(self._lib.uvo_extract_3d_points(self._h, _p(k1), _p(k2), n, *[_p(x) for x in a], _p(p4), _p(pts), _p(idx), C.byref(g)))
        return pts[:g.value].copy(), idx[:g.value].copy()

    def reproject_errors(self, world_points, R, t, K, img_points):
        w, img = _np(world_points, np.float64), _np(img_points, np.float32)
        R, t, K = _np(R, np.float64), _np(t, np.float64), _np(K, np.float64)
        n = len(w)
        err = np.empty(max(n, 1))
        self._check(self._lib.uvo_reproject_errors(self._h, _p(w), n, _p(R), _p(t), _p(K), _p(img), _p(err)))
        return err[:n].copy()

    def solvePnPRansac(self, object_points, image_points, K, iterations_count=None, reprojection_error=None, confidence=None):
        p = self.params
        it = int(p.ITERATIONS_COUNT if iterations_count is None else iterations_count)
        re_ = float(p.REPROJECTION_ERROR_THRESHOLD if reprojection_error is None else reprojection_error)
        cf = float(p.CONFIDENCE if confidence is None else confidence)
        obj, img, K = _np(object_points, np.float64), _np(image_points, np.float32), _np(K, np.float64)
        n = len(obj)
        rvec, tvec = np.zeros(3), np.zeros(3)
        inl = np.empty(max(n, 1), np.int32)
        ni, ok = C.c_int(0), C.c_int(0)
        self._check(self._lib.uvo_solve_pnp_ransac(self._h, _p(obj), _p(img), n, _p(K), it, C.c_float(re_), C.c_double(cf),
                                                   _p(rvec), _p(tvec), _p(inl), C.byref(ni), C.byref(ok)))
        return bool(ok.value), rvec, tvec, inl[:ni.value].copy()

    def Rodrigues(self, x):
        x = _np(x, np.float64).ravel()
        out = np.empty(9 if x.size == 3 else 3)
        st = self._lib.uvo_rodrigues(_p(x), int(x.size), _p(out))
        if st != 0:
            raise UvoError(st, "uvo_rodrigues: input must have 3 or 9 elements")
        return out.reshape(3, 3) if x.size == 3 else out

    # ------------------------------------------------------------------ stereo loop (visual_odometry.h:406-741)
    def stereo_set_rig(self, K_left, K_right, R_right, t_right):
        a = [_np(x, np.float64) for x in (K_left, K_right, R_right, t_right)]
        self._check(self._lib.uvo_stereo_set_rig(self._h, *[_p(x) for x in a]))
        self._drained()

    def stereo_reset(self):
        self._check(self._lib.uvo_stereo_reset(self._h))
        self._drained()

    def stereo_step(self, left, right, dt: float = 0.05) -> StereoResult:
        h, w = left.shape[-2], left.shape[-1]
        pl, ml, kl = _ptr_mem(left, np.uint8)
        pr, mr, kr = _ptr_mem(right, np.uint8)
        if ml != mr:
            raise ValueError("left and right must live in the same memory space")
        r = StereoResult()
        self._order_after_producer(left, right)
        self._check(self._lib.uvo_stereo_step(self._h, pl, pr, w, h, w, ml, C.c_double(dt), C.byref(r)))
        return r

    def get_image(self, rgb, desired_width, K, dist4, newK, clahe=True, clip_limit=3, device_out=False):
        """get_image (VO_utility.cpp:337-379): resize INTER_AREA -> RGB2GRAY -> undistort -> optional CLAHE.
        rgb: HxWx3 uint8 numpy array, or a CUDA torch tensor of that shape.  Returns a numpy array, or a CUDA tensor
        when device_out is set."""
        K, dist4, newK = _np(K, np.float64), _np(dist4, np.float64), _np(newK, np.float64)
        if hasattr(rgb, "data_ptr"):
            h, w, _ = rgb.shape
            src, mem, stride = C.c_void_p(rgb.data_ptr()), 1, int(rgb.stride(0))
        else:
            rgb = _np(rgb, np.uint8); h, w, _ = rgb.shape
            src, mem, stride = _p(rgb), 0, w * 3
        dh = int(h / (w / desired_width))
        ow, oh = C.c_int(0), C.c_int(0)
        if device_out:
            import torch
            out = torch.empty((dh, desired_width), dtype=torch.uint8, device="cuda")
            dst, omem = C.c_void_p(out.data_ptr()), 1
        else:
            out = np.empty((dh, desired_width), np.uint8)
            dst, omem = _p(out), 0
        self._check(self._lib.uvo_get_image(self._h, src, w, h, stride, mem, _p(K), _p(dist4), _p(newK), int(desired_width),
                                            int(bool(clahe)), int(clip_limit), dst, omem, C.byref(ow), C.byref(oh)))
        assert (ow.value, oh.value) == (desired_width, dh)
        return out

    def decode_image(self, data: bytes, fmt: str = "bgr8; jpeg compressed bgr8", device_out=False):
        """from_ros_to_cv_image (math_utility.cpp:154-173): the payload of a sensor_msgs/CompressedImage -> H x W x 3 BGR (or H x W grey);
        a format containing "bayer" is demosaiced with COLOR_BayerBGGR2BGR.  Returns a numpy array, or a CUDA tensor when device_out."""
        buf = np.frombuffer(data, np.uint8)
        w, h, ch = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.uvo_decode_image(self._h, _p(buf), len(buf), fmt.encode(), None, 0, 0, C.byref(w), C.byref(h), C.byref(ch)))
        shape = (h.value, w.value, ch.value) if ch.value > 1 else (h.value, w.value)
        if device_out:
            import torch
            out = torch.empty(shape, dtype=torch.uint8, device="cuda")
            dst, omem, nb = C.c_void_p(out.data_ptr()), 1, out.numel()
        else:
            out = np.empty(shape, np.uint8)
            dst, omem, nb = _p(out), 0, out.nbytes
        self._check(self._lib.uvo_decode_image(self._h, _p(buf), len(buf), fmt.encode(), dst, nb, omem, C.byref(w), C.byref(h), C.byref(ch)))
        return out

    def bayer_bggr2bgr(self, bayer):
        h, w = bayer.shape
        p, mem, keep = _ptr_mem(bayer, np.uint8)
        out = np.empty((h, w, 3), np.uint8)
        self._check(self._lib.uvo_bayer_bggr2bgr(self._h, p, w, h, w, mem, _p(out), 0))
        return out

    def stereo_set_depth(self, depth):
        """Number of consecutive pairs that may be in flight between stereo_submit and stereo_collect (default 2)."""
        self._check(self._lib.uvo_stereo_set_depth(self._h, int(depth)))
        self._drained()

    def stereo_set_batch(self, pairs: int):
        """uvo_stereo_set_batch: 2 = consecutive pairs are queued two at a time, one launch per kernel for both (batch consumers)."""
        self._check(self._lib.uvo_stereo_set_batch(self._h, int(pairs)))

    def stereo_submit(self, left, right):
        """Enqueue detect..extract_3Dpoints of a pair (no host sync); at most two pairs in flight."""
        h, w = left.shape[-2], left.shape[-1]
        pl, ml, kl = _ptr_mem(left, np.uint8)
        pr, mr, kr = _ptr_mem(right, np.uint8)
        if ml != mr:
            raise ValueError("left and right must live in the same memory space")
        self._order_after_producer(left, right)
        self._check(self._lib.uvo_stereo_submit(self._h, pl, pr, w, h, w, ml))
        self._inflight.append((kl, kr))             # read asynchronously: alive until the pair is collected

    def stereo_collect(self, dt: float = 0.05, out: StereoResult | None = None) -> StereoResult:
        """Finish the oldest submitted pair (PnP-RANSAC + pose); same result as stereo_step.  `out`: the StereoResult to fill
        (e.g. an element of a preallocated ctypes array, so that a long loop copies nothing on the Python side)."""
        r = StereoResult() if out is None else out
        before = self._lib.uvo_ctx_pending(self._h)
        try:
            self._check(self._lib.uvo_stereo_collect(self._h, C.c_double(dt), C.byref(r)))
        finally:
            self._release_collected(before)
        return r

    def stereo_get(self, what: str):
        spec = {"kps_left": KP_DTYPE, "kps_right": KP_DTYPE, "desc_left": np.dtype(("f4", 128 if (self.params.SURF_EXTENDED or getattr(self, "_feature_sift", False)) else 64)),
                "desc_right": np.dtype(("f4", 128 if (self.params.SURF_EXTENDED or getattr(self, "_feature_sift", False)) else 64)), "matches_stereo": DM_DTYPE, "matches_tri": DM_DTYPE,
                "points4d": np.dtype(("f4", 4)), "good_pts": np.dtype(("f8", 3)), "good_idx": np.dtype("i4"),
                "inliers": np.dtype("i4")}[what]
        buf = np.zeros(self.max_kpts, spec)
        n = self._lib.uvo_stereo_get(self._h, what.encode(), _p(buf), buf.nbytes)
        if n < 0:
            raise UvoError(3, "stereo_get buffer too small")
        out = buf[:n].copy()
        if what == "points4d":
            out = np.ascontiguousarray(out.T)       # 4 x T like cv::triangulatePoints
        return out

    # ------------------------------------------------------------------ mono relative pose (VO_utility.cpp:134-180)
    def findEssentialMat(self, pts1, pts2, K, method=8, prob=0.99, threshold=1.0, max_iters=1000):
        p1, p2, K = _np(pts1, np.float32), _np(pts2, np.float32), _np(K, np.float64)
        n = len(p1)
        E = np.zeros((3, 3)); mask = np.zeros(max(n, 1), np.uint8); ok = C.c_int(0)
        self._check(self._lib.uvo_find_essential_mat(self._h, _p(p1), _p(p2), n, _p(K), int(method), C.c_double(prob), C.c_double(threshold),
                                                     int(max_iters), _p(E), _p(mask), C.byref(ok)))
        return bool(ok.value), E, mask[:n].copy()

    def recoverPose(self, E, pts1, pts2, K, mask):
        E, p1, p2, K = _np(E, np.float64), _np(pts1, np.float32), _np(pts2, np.float32), _np(K, np.float64)
        m = _np(mask, np.uint8).copy(); R = np.empty((3, 3)); t = np.empty(3); good = C.c_int(0)
        self._check(self._lib.uvo_recover_pose(self._h, _p(E), _p(p1), _p(p2), len(p1), _p(K), _p(R), _p(t), _p(m), C.byref(good)))
        return good.value, R, t, m

    def findHomography(self, pts1, pts2, method=8, threshold=3.0, max_iters=2000, confidence=0.995):
        p1, p2 = _np(pts1, np.float32), _np(pts2, np.float32)
        n = len(p1)
        H = np.zeros((3, 3)); mask = np.zeros(max(n, 1), np.uint8); ok = C.c_int(0)
        self._check(self._lib.uvo_find_homography(self._h, _p(p1), _p(p2), n, int(method), C.c_double(threshold), int(max_iters),
                                                  C.c_double(confidence), _p(H), _p(mask), C.byref(ok)))
        return bool(ok.value), H, mask[:n].copy()

    def decomposeHomographyMat(self, H, K):
        H, K = _np(H, np.float64), _np(K, np.float64)
        Rs = np.empty((4, 3, 3)); ts = np.empty((4, 3)); ns = np.empty((4, 3)); n = C.c_int(0)
        st = self._lib.uvo_decompose_homography_mat(_p(H), _p(K), _p(Rs), _p(ts), _p(ns), C.byref(n))
        if st != 0:
            raise UvoError(st, "uvo_decompose_homography_mat")
        return Rs[:n.value].copy(), ts[:n.value].copy(), ns[:n.value].copy()

    def recover_pose_homography(self, H, pts1, pts2, K):
        H, p1, p2, K = _np(H, np.float64), _np(pts1, np.float32), _np(pts2, np.float32), _np(K, np.float64)
        R = np.full((3, 3), np.nan); t = np.full(3, np.nan); g = C.c_int(0)
        self._check(self._lib.uvo_recover_pose_homography(self._h, _p(H), _p(p1), _p(p2), len(p1), _p(K), _p(R), _p(t), C.byref(g)))
        return g.value, R, t

    def select_estimation_method(self, pts1, pts2, distance=None) -> bool:
        p1, p2 = _np(pts1, np.float32), _np(pts2, np.float32)
        d = int(self.params.DISTANCE if distance is None else distance)
        r = self._lib.uvo_select_estimation_method(_p(p1), _p(p2), len(p1), d)
        if r < 0:
            raise MemoryError("uvo_select_estimation_method: no host memory for the median's scratch")
        return bool(r)

    def estimate_relative_pose(self, pts1, pts2, K, use_essential=True, R0=None, t0=None):
        p1, p2, K = _np(pts1, np.float32), _np(pts2, np.float32), _np(K, np.float64)
        n = len(p1)
        R = np.eye(3) if R0 is None else _np(R0, np.float64).copy()
        t = np.zeros(3) if t0 is None else _np(t0, np.float64).copy()
        in1 = np.empty((max(n, 1), 2), np.float32); in2 = np.empty((max(n, 1), 2), np.float32)
        ue, nin, succ = C.c_int(int(use_essential)), C.c_int(0), C.c_int(0)
        mask = np.zeros(max(n, 1), np.uint8)
        self._check(self._lib.uvo_estimate_relative_pose(self._h, _p(p1), _p(p2), n, _p(K), C.byref(ue), _p(R), _p(t), _p(in1), _p(in2),
                                                         C.byref(nin), _p(mask), C.byref(succ)))
        return bool(succ.value), bool(ue.value), R, t, in1[:nin.value].copy(), in2[:nin.value].copy(), mask[:n].copy()

    # ------------------------------------------------------------------ mono loop (visual_odometry.h:167-398)
    def mono_set_camera(self, K):
        K = _np(K, np.float64)
        self._check(self._lib.uvo_mono_set_camera(self._h, _p(K)))
        self._drained()

    def mono_step(self, img, range_=1.0, dt: float = 0.05) -> MonoResult:
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        r = MonoResult()
        self._order_after_producer(img)
        self._check(self._lib.uvo_mono_step(self._h, p, w, h, w, mem, C.c_double(range_), C.c_double(dt), C.byref(r)))
        return r

    def mono_submit(self, img, range_=1.0):
        """Pipelined mono frame (uvo_mono_submit): at most `stereo_set_depth` frames in flight, collect in order."""
        h, w = img.shape[-2], img.shape[-1]
        p, mem, keep = _ptr_mem(img, np.uint8)
        self._order_after_producer(img)
        self._check(self._lib.uvo_mono_submit(self._h, p, w, h, w, mem, C.c_double(range_)))
        self._inflight.append((keep,))

    def mono_collect(self, dt: float = 0.05) -> MonoResult:
        r = MonoResult()
        before = self._lib.uvo_ctx_pending(self._h)
        try:
            self._check(self._lib.uvo_mono_collect(self._h, C.c_double(dt), C.byref(r)))
        finally:
            self._release_collected(before)
        return r

    def mono_reset(self):
        self._check(self._lib.uvo_mono_reset(self._h))
        self._drained()

    def mono_get(self, what: str):
        spec = {"kps": KP_DTYPE, "matches": DM_DTYPE, "mask": np.dtype("u1"), "good_pts": np.dtype(("f8", 3))}[what]
        buf = np.zeros(self.max_kpts, spec)
        n = self._lib.uvo_mono_get(self._h, what.encode(), _p(buf), buf.nbytes)
        return buf[:max(n, 0)].copy()

    # ------------------------------------------------------------------ timing
    def timing_enable(self, on: bool = True):
        self._check(self._lib.uvo_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._check(self._lib.uvo_timing_reset(self._h))

    def timing(self) -> dict:
        out = {}
        for i in range(self._lib.uvo_timing_count(self._h)):
            ms, n = C.c_double(0), C.c_longlong(0)
            self._lib.uvo_timing_get(self._h, i, C.byref(ms), C.byref(n))
            out[self._lib.uvo_timing_name(self._h, i).decode()] = (ms.value, n.value)
        return out

    # ------------------------------------------------------------------ pipeline trace
    def trace_enable(self, on: bool = True):
        """Record device and host timestamps of every pipelined pair's phases (uvo_trace_enable; clears the ring)."""
        self._check(self._lib.uvo_trace_enable(self._h, int(on)))

    def trace_read(self):
        """-> structured array (TRACE_DTYPE) of the traced pairs, oldest first (uvo_trace_read; nothing may be in flight)."""
        buf = np.zeros(256 * 16, TRACE_DTYPE)
        n = self._lib.uvo_trace_read(self._h, _p(buf), len(buf))
        if n < 0:
            raise UvoError(1, "uvo_trace_read: pairs in flight")
        return buf[:min(n, len(buf))].copy()


def resize_camera_matrix(original_width: int, original_height: int, desired_width: int, K, dist4):
    """resize_camera_matrix (VO_utility.cpp:658-675): -> (K scaled by the width ratio, newK = getOptimalNewCameraMatrix(alpha=0),
    desired_height).  Host arithmetic, once per run; no context needed."""
    Ks = np.array(K, np.float64).reshape(3, 3).copy()
    newK = np.zeros((3, 3), np.float64)
    d4 = _np(dist4, np.float64)
    dh = C.c_int(0)
    st = _lib.lib().uvo_resize_camera_matrix(int(original_width), int(original_height), int(desired_width), _p(Ks), _p(d4), _p(newK), C.byref(dh))
    if st != 0:
        raise UvoError(st, "uvo_resize_camera_matrix")
    return Ks, newK, dh.value
