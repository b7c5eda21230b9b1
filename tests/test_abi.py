"""The C-ABI library loads without a GPU, exports every symbol include/uvo_hip.h declares, keeps the
reference's POD layouts and parameter defaults, and refuses to compute without a device (there is no
CPU fallback in the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "uvo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(uvo_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from ergo_uvo_amd import _lib
    lib = _lib.lib()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/uvo_hip.h but not exported by libuvo_hip.so"
    assert set(_lib.EXPORTS) == set(names)


def test_pod_layouts_match_opencv_types():
    import ergo_uvo_amd as uvo
    assert uvo.KP_DTYPE.itemsize == 28 and uvo.DM_DTYPE.itemsize == 16          # cv::KeyPoint, cv::DMatch
    assert [uvo.KP_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]
    assert [uvo.DM_DTYPE.fields[f][1] for f in ("queryIdx", "trainIdx", "imgIdx", "distance")] == [0, 4, 8, 12]


def test_parameter_defaults_are_the_shipped_yaml_values():
    import ergo_uvo_amd as uvo
    s = uvo.Params.stereo()      # uvo/config/stereo_VO_parameters.yaml
    assert (s.LOWE_RATIO_THRESHOLD, s.REPROJECTION_TOLERANCE, s.MIN_NUM_FEATURES, s.MIN_NUM_3DPOINTS, s.MIN_NUM_INLIERS) == (0.8, 3.0, 5, 5, 5)
    assert (s.ITERATIONS_COUNT, s.REPROJECTION_ERROR_THRESHOLD, s.CONFIDENCE, s.USE_EXTRINSIC_GUESS, s.PNP_METHOD_FLAG) == (1000, 1.0, 0.99, 0, 1)
    assert (s.SURF_MIN_HESSIAN, s.SURF_OCTAVES_NUMBER, s.SURF_OCTAVES_LAYERS, s.SURF_EXTENDED, s.SURF_UPRIGHT) == (1500, 4, 3, 0, 1)
    assert s.ESSENTIAL_OUTLIER_METHOD == 0 and s.DISTANCE == 0                # keys absent from the stereo yaml stay zero
    m = uvo.Params.mono()        # uvo/config/mono_VO_parameters.yaml
    assert (m.DISTANCE, m.LOWE_RATIO_THRESHOLD, m.ESSENTIAL_OUTLIER_METHOD, m.HOMOGRAPHY_OUTLIER_METHOD) == (10, 0.7, 4, 4)
    assert (m.ESSENTIAL_MAX_ITERS, m.ESSENTIAL_CONFIDENCE, m.ESSENTIAL_THRESHOLD) == (2000, 0.99, 0.1)
    assert (m.HOMOGRAPHY_DISTANCE, m.VPF_THRESHOLD, m.REPROJECTION_TOLERANCE) == (50.0, 0.4, 0.1)
    assert (m.MIN_NUM_FEATURES, m.MIN_NUM_INLIERS, m.MIN_NUM_3DPOINTS, m.SURF_MIN_HESSIAN) == (20, 10, 5, 50)


def test_rodrigues_host_entry_point():
    from ergo_uvo_amd import _lib
    lib = _lib.lib()
    r = np.array([0.3, -0.2, 0.1]); R = np.empty(9); back = np.empty(3)
    assert lib.uvo_rodrigues(r.ctypes.data_as(C.c_void_p), 3, R.ctypes.data_as(C.c_void_p)) == 0
    Rm = R.reshape(3, 3)
    assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(Rm) - 1) < 1e-14
    assert lib.uvo_rodrigues(R.ctypes.data_as(C.c_void_p), 9, back.ctypes.data_as(C.c_void_p)) == 0
    assert np.allclose(back, r, atol=1e-13)
    assert lib.uvo_rodrigues(r.ctypes.data_as(C.c_void_p), 4, R.ctypes.data_as(C.c_void_p)) == 1      # UVO_INVALID_ARG


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ergo_uvo_amd as uvo
    with pytest.raises(uvo.UvoError) as e:
        uvo.Context(uvo.Params.stereo(), 0, 640, 480, 1024)
    assert e.value.status in (4, 5)          # UVO_HIP_ERROR / UVO_NO_DEVICE: the product path fails loudly


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ergo_uvo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("# no oracle", ""), f"{f} mentions the oracle"
