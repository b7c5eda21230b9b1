"""Loader for libuvo_hip.so (the C ABI declared in include/uvo_hip.h).

There is no CPU fallback: if the HIP library is missing it is (re)built with hipcc, and if that
fails -- or no GPU is present when a context is created -- the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("UVO_HIP_LIB") or os.path.join(_HERE, "lib", "libuvo_hip.so")      # UVO_HIP_LIB: another build of the same ABI (A/B measurements)

EXPORTS = [
    "uvo_params_default_stereo", "uvo_params_default_mono", "uvo_ctx_create", "uvo_ctx_destroy", "uvo_last_error",
    "uvo_ctx_stream", "uvo_ctx_set_params", "uvo_ctx_set_feature_detector", "uvo_ctx_set_producer_stream", "uvo_ctx_warning", "uvo_ctx_pending", "uvo_ctx_host_policy", "uvo_surf_detect", "uvo_sift_detect", "uvo_sift_layer", "uvo_akaze_detect", "uvo_akaze_plane", "uvo_orb_configure", "uvo_orb_set_pattern", "uvo_orb_detect", "uvo_orb_plane", "uvo_integral", "uvo_hessian_layer",
    "uvo_match_knn2_ratio", "uvo_match_knn2", "uvo_match_knn2_ratio_dim", "uvo_match_knn2_dim", "uvo_match_knn2_ratio_hamming", "uvo_match_knn2_hamming", "uvo_triangulate_points", "uvo_extract_3d_points",
    "uvo_solve_pnp_ransac", "uvo_reproject_errors", "uvo_rodrigues", "uvo_stereo_set_rig", "uvo_stereo_reset", "uvo_stereo_step",
    "uvo_stereo_set_depth", "uvo_stereo_submit", "uvo_stereo_collect",
    "uvo_stereo_get", "uvo_find_essential_mat", "uvo_recover_pose", "uvo_find_homography", "uvo_decompose_homography_mat",
    "uvo_recover_pose_homography", "uvo_select_estimation_method", "uvo_estimate_relative_pose", "uvo_mono_set_camera",
    "uvo_mono_reset", "uvo_mono_step", "uvo_mono_submit", "uvo_mono_collect", "uvo_mono_get", "uvo_get_image", "uvo_decode_image", "uvo_bayer_bggr2bgr", "uvo_resize_camera_matrix", "uvo_timing_enable", "uvo_timing_count", "uvo_timing_name", "uvo_timing_get", "uvo_timing_reset", "uvo_trace_enable", "uvo_trace_read",
]


# exported, but not part of the drop-in ABI (ergo_uvo_amd/csrc/uvo_experimental.h): measurement hooks
EXPERIMENTAL = ["uvo_stereo_set_batch"]


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into lib/libuvo_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "uvo_hip.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        # PyTorch-ROCm bundles its own HIP runtime; if /opt/rocm's is initialised first (by this library), torch later reports
        # "No HIP GPUs are available".  Loading torch first makes both use the same runtime, in either order of use.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.uvo_last_error.restype = C.c_char_p
        _lib.uvo_last_error.argtypes = [C.c_void_p]
        _lib.uvo_ctx_stream.restype = C.c_void_p
        _lib.uvo_ctx_stream.argtypes = [C.c_void_p]
        _lib.uvo_ctx_warning.restype = C.c_char_p
        _lib.uvo_ctx_warning.argtypes = [C.c_void_p]
        _lib.uvo_ctx_host_policy.restype = C.c_char_p
        _lib.uvo_ctx_host_policy.argtypes = [C.c_void_p]
        _lib.uvo_ctx_set_producer_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.uvo_decode_image.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.uvo_sift_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.uvo_ctx_set_feature_detector.argtypes = [C.c_void_p, C.c_char_p]
        _lib.uvo_ctx_pending.restype = C.c_int
        _lib.uvo_ctx_pending.argtypes = [C.c_void_p]
        _lib.uvo_ctx_destroy.argtypes = [C.c_void_p]
        _lib.uvo_ctx_destroy.restype = None
        _lib.uvo_timing_name.restype = C.c_char_p
        _lib.uvo_timing_name.argtypes = [C.c_void_p, C.c_int]
        _lib.uvo_trace_read.restype = C.c_int
        _lib.uvo_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.uvo_trace_enable.argtypes = [C.c_void_p, C.c_int]
        for name in EXPORTS + EXPERIMENTAL:
            getattr(_lib, name)  # fail loudly if the ABI and the header drift apart
    return _lib
