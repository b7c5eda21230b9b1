"""The C++ host-side mirror of the reference's uvo_libraries API (include/uvo_libraries_hip/, ergo_uvo_amd/shim/).

CPU: the shim library builds, exports every function the reference's VO_utility.h declares for the path, and its
header compiles stand-alone.  GPU: a C++ driver runs the stereo node's loop (visual_odometry.h:474-739) through the
shim and every per-frame record must equal the CPU oracle's state machine bit for bit."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "ergo_uvo_amd", "shim")
SHIM_LIB = os.path.join(ROOT, "ergo_uvo_amd", "lib", "libuvo_libraries_hip.so")
DRIVER = os.path.join(ROOT, "tests", "cpp", "build", "shim_stereo_node")

# VO_utility.h:96-117, minus ROS parameter readers, resize_camera_matrix (one-time setup) and show_matches
REFERENCE_FUNCTIONS = [
    "compute_projection_matrix", "compute_scale_factor", "convert_3Dpoints_camera", "convert_from_homogeneous_coords",
    "detect_features", "get_image", "estimate_relative_pose", "extract_3Dpoints", "extract_3Dpoints_and_reprojection", "extract_inliers",
    "match_features", "recover_pose_homography", "reproject_errors", "select_desired_descriptors", "select_desired_keypoints",
    "select_estimation_method",
]
REFERENCE_GLOBALS = [
    "FEATURE_DETECTOR", "DESIRED_WIDTH", "CLAHE_CORRECTION", "CLIP_LIMIT", "DISTANCE", "ESSENTIAL_OUTLIER_METHOD", "ESSENTIAL_MAX_ITERS", "ESSENTIAL_CONFIDENCE", "ESSENTIAL_THRESHOLD",
    "HOMOGRAPHY_OUTLIER_METHOD", "HOMOGRAPHY_MAX_ITERS", "HOMOGRAPHY_CONFIDENCE", "HOMOGRAPHY_THRESHOLD", "HOMOGRAPHY_DISTANCE",
    "VPF_THRESHOLD", "REPROJECTION_TOLERANCE", "LOWE_RATIO_THRESHOLD", "MIN_NUM_FEATURES", "MIN_NUM_3DPOINTS", "MIN_NUM_INLIERS",
    "ITERATIONS_COUNT", "REPROJECTION_ERROR_THRESHOLD", "CONFIDENCE", "USE_EXTRINSIC_GUESS", "PNP_METHOD_FLAG",
    "SURF_MIN_HESSIAN", "SURF_OCTAVES_NUMBER", "SURF_OCTAVES_LAYERS", "SURF_EXTENDED", "SURF_UPRIGHT", "use_essential",
]


def _build():
    from ergo_uvo_amd import _lib
    _lib.build()
    subprocess.check_call(["make", "-C", SHIM_DIR, "-s"])


def test_shim_builds_and_exports_reference_surface():
    _build()
    assert os.path.exists(SHIM_LIB) and os.path.exists(DRIVER)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", SHIM_LIB], text=True)
    for f in REFERENCE_FUNCTIONS:
        assert any(line.split(" ", 2)[-1].startswith(f + "(") for line in syms.splitlines()), f
    assert sum(1 for line in syms.splitlines() if line.split(" ", 2)[-1].startswith("match_features(")) == 2   # both overloads
    for g in REFERENCE_GLOBALS:
        assert any(line.split()[-1] == g or line.split(" ", 2)[-1].startswith(g + "[") for line in syms.splitlines()), g
    for f in ("uvo_hip::triangulatePoints(", "uvo_hip::solvePnPRansac(", "uvo_hip::Rodrigues(", "uvo_hip::configure("):
        assert f in syms, f
    # the shim is marshalling only: it must not link or name the test oracle
    blob = open(SHIM_LIB, "rb").read()
    assert b"orc_" not in blob and b"liboracle" not in blob


def test_shim_header_is_self_contained(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "uvo_libraries_hip/VO_utility_hip.h"\nint main() { uvocv::Mat m(2, 2, uvocv::CV_64FC1); return m.rows - 2; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-DUVO_NO_OPENCV", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"), str(src)])


def test_cv_compat_mat_semantics(tmp_path):
    """push_back / row / t / clone of the stand-in Mat behave like cv::Mat for the cases the API relies on."""
    src = tmp_path / "m.cpp"
    src.write_text(r'''
#include "uvo_libraries_hip/cv_compat.h"
#include <cstdio>
using namespace uvocv;
int main() {
    Mat idx; for (int i = 0; i < 5; i++) idx.push_back(i * 3);
    if (idx.rows != 5 || idx.cols != 1 || idx.type() != CV_32SC1 || idx.at<int>(4, 0) != 12) return 1;
    Mat a(2, 3, CV_64FC1); for (int i = 0; i < 6; i++) a.at<double>(i / 3, i % 3) = i;
    Mat alias = a; Mat b = a.clone(); a.push_back(b);                 // append must not disturb an aliasing header
    if (a.rows != 4 || alias.rows != 2 || a.at<double>(3, 2) != 5 || alias.at<double>(1, 2) != 5) return 2;
    Mat t = b.t(); if (t.rows != 3 || t.cols != 2 || t.at<double>(2, 1) != 5 || t.at<double>(1, 0) != 1) return 3;
    Mat r = b.row(1); if (r.rows != 1 || r.at<double>(0, 0) != 3) return 4;
    Mat e = Mat::eye(3, 3, CV_64FC1); if (e.at<double>(1, 1) != 1 || e.at<double>(1, 2) != 0) return 5;
    if (sizeof(KeyPoint) != 28 || sizeof(DMatch) != 16 || sizeof(Point2f) != 8) return 6;
    return 0;
}
''')
    exe = tmp_path / "m"
    subprocess.check_call(["g++", "-std=c++17", "-DUVO_NO_OPENCV", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


@pytest.mark.gpu
def test_stereo_node_loop_through_the_shim_matches_oracle(oracle, scene_small, tmp_path):
    from ergo_uvo_amd import synth
    _build()
    rig = synth.stereo_rig(640)
    seq = [scene_small[k] for k in (0, 1, 2, 1, 0)]
    H, W = seq[0][0].shape
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<4i", W, H, len(seq), 1500))
        for m in (rig.K_left, rig.K_right, rig.R_right, rig.t_right):
            f.write(np.ascontiguousarray(m, np.float64).tobytes())
        for L, R in seq:
            f.write(np.ascontiguousarray(L).tobytes()); f.write(np.ascontiguousarray(R).tobytes())
    res = subprocess.run([DRIVER, str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    rec = np.fromfile(outp, np.dtype([("i", "<i4", 8), ("d", "<f8", 9)]))
    assert len(rec) == len(seq)
    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    nvalid = 0
    for k, (L, R) in enumerate(seq):
        o = ovo.step(L, R, 0.05)
        want = [o.valid, o.initialized, o.n_left, o.n_right, o.n_stereo_matches, o.n_tri_matches, o.n_good3d, o.n_inliers]
        assert list(rec["i"][k]) == want, (k, list(rec["i"][k]), want)
        d = np.array(list(o.rvec) + list(o.tvec) + list(o.t_prev_curr))
        for a, b in zip(rec["d"][k].reshape(3, 3), d.reshape(3, 3)):          # pose: north_star tolerance (parallel refit, uvo_epnp_fast.h)
            assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), (k, rec["d"][k], d)
        nvalid += o.valid
    assert nvalid == len(seq) - 1


@pytest.mark.gpu
def test_stereo_node_loop_on_sift_features_through_the_shim_matches_oracle(oracle, scene_small, tmp_path):
    """FEATURE_DETECTOR = "SIFT" (the reference's global): the same node loop, detect_features and match_features taking their SIFT
    branches (VO_utility.cpp:107-112, 525-529), against the oracle's state machine switched to the same detector."""
    from ergo_uvo_amd import synth
    _build()
    rig = synth.stereo_rig(640)
    seq = [scene_small[k] for k in (0, 1, 2, 1)]
    H, W = seq[0][0].shape
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<4i", W, H, len(seq), -1))
        for m in (rig.K_left, rig.K_right, rig.R_right, rig.t_right):
            f.write(np.ascontiguousarray(m, np.float64).tobytes())
        for L, R in seq:
            f.write(np.ascontiguousarray(L).tobytes()); f.write(np.ascontiguousarray(R).tobytes())
    res = subprocess.run([DRIVER, str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    rec = np.fromfile(outp, np.dtype([("i", "<i4", 8), ("d", "<f8", 9)]))
    assert len(rec) == len(seq)
    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    ovo.use_sift()
    nvalid = 0
    for k, (L, R) in enumerate(seq):
        o = ovo.step(L, R, 0.05)
        want = [o.valid, o.initialized, o.n_left, o.n_right, o.n_stereo_matches, o.n_tri_matches, o.n_good3d, o.n_inliers]
        assert list(rec["i"][k]) == want, (k, list(rec["i"][k]), want)
        d = np.array(list(o.rvec) + list(o.tvec) + list(o.t_prev_curr))
        for a, b in zip(rec["d"][k].reshape(3, 3), d.reshape(3, 3)):
            assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12), (k, rec["d"][k], d)
        nvalid += o.valid
    assert nvalid == len(seq) - 1 and rec["i"][-1][2] > 1000          # valid poses from > 1000 SIFT keypoints per image


@pytest.mark.gpu
@pytest.mark.parametrize("detector,code", [("AKAZE", -2), ("ORB", -3)])
def test_stereo_node_loop_on_binary_features_through_the_shim_matches_oracle(oracle, scene_small, tmp_path, detector, code):
    """FEATURE_DETECTOR = "AKAZE" / "ORB": the same node loop on binary descriptors -- detect_features' AKAZE / ORB branches
    (VO_utility.cpp:93-105), match_features' Hamming branch (VO_utility.cpp:520-524), select_desired_descriptors on CV_8U rows
    (VO_utility.cpp:683-696 copies rows of any type) -- against the oracle's state machine switched to the same detector.  ORB's sampling
    table (an input: OpenCV's bit_pattern_31_ cannot be restated) is the one OpenCV's makeRandomPattern draws, read by the C++ surface from
    UVO_ORB_PATTERN_FILE."""
    from ergo_uvo_amd import synth
    _build()
    rig = synth.stereo_rig(640)
    seq = [scene_small[k] for k in (0, 1, 2, 1)]
    H, W = seq[0][0].shape
    inp, outp, patf = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "bit_pattern_31.txt"
    pat = oracle.orb_random_pattern()
    patf.write_text(",\n".join(", ".join(str(int(v)) for v in row) for row in pat.reshape(256, 4)) + "\n")
    with open(inp, "wb") as f:
        f.write(struct.pack("<4i", W, H, len(seq), code))
        for m in (rig.K_left, rig.K_right, rig.R_right, rig.t_right):
            f.write(np.ascontiguousarray(m, np.float64).tobytes())
        for L, R in seq:
            f.write(np.ascontiguousarray(L).tobytes()); f.write(np.ascontiguousarray(R).tobytes())
    res = subprocess.run([DRIVER, str(inp), str(outp)], capture_output=True, text=True, timeout=300, env=dict(os.environ, UVO_ORB_PATTERN_FILE=str(patf)))
    assert res.returncode == 0, res.stderr
    rec = np.fromfile(outp, np.dtype([("i", "<i4", 8), ("d", "<f8", 9)]))
    assert len(rec) == len(seq)
    ovo = oracle.StereoVO(oracle.stereo_params(1500), rig.K_left, rig.K_right, rig.R_right, rig.t_right, max_kpts=16384)
    ovo.use_detector(detector, pat)
    nvalid = 0
    for k, (L, R) in enumerate(seq):
        o = ovo.step(L, R, 0.05)
        want = [o.valid, o.initialized, o.n_left, o.n_right, o.n_stereo_matches, o.n_tri_matches, o.n_good3d, o.n_inliers]
        assert list(rec["i"][k]) == want, (k, list(rec["i"][k]), want)
        d = np.array(list(o.rvec) + list(o.tvec) + list(o.t_prev_curr))
        for a, b in zip(rec["d"][k].reshape(3, 3), d.reshape(3, 3)):
            assert np.linalg.norm(a - b) <= 1e-4 * max(np.linalg.norm(b), 1e-12), (k, rec["d"][k], d)
        nvalid += o.valid
    assert nvalid == len(seq) - 1 and rec["i"][-1][7] > (500 if detector == "AKAZE" else 1500)      # valid poses, hundreds of PnP inliers


@pytest.mark.gpu
def test_get_image_through_the_shim_matches_oracle(oracle, tmp_path):
    _build()
    rng = np.random.default_rng(9)
    H, W, DW = 180, 320, 240
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.clip((120 + 90 * np.sin(xx / 11.0) * np.cos(yy / 6.0))[..., None] + rng.normal(0, 10, (H, W, 3)), 0, 255).astype(np.uint8)
    dh = int(H / (W / DW))
    K = np.array([[200.0, 0, 121.0], [0, 205.0, 66.0], [0, 0, 1.0]])
    newK = np.array([[190.0, 0, 120.0], [0, 195.0, 67.5], [0, 0, 1.0]])
    dist = np.array([-0.25, 0.07, 0.001, -0.002])
    inp, outp = tmp_path / "gi.bin", tmp_path / "go.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<5i", W, H, DW, 1, 8))
        for m in (K, dist, newK):
            f.write(np.ascontiguousarray(m, np.float64).tobytes())
        f.write(img.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_get_image"), str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = open(outp, "rb").read()
    ow, oh = struct.unpack("<2i", raw[:8])
    assert (ow, oh) == (DW, dh)
    got = np.frombuffer(raw[8:8 + ow * oh], np.uint8).reshape(oh, ow)
    assert np.array_equal(got, oracle.get_image(img, DW, K, dist, newK, True, 8))
    # resize_camera_matrix through the same function surface: K scaled in place, newK = getOptimalNewCameraMatrix(alpha = 0)
    both = np.frombuffer(raw[8 + ow * oh:], np.float64)
    oK, oN, odh = oracle.resize_camera_matrix(img.shape[1], img.shape[0], DW, K, dist)
    assert odh == dh
    assert np.array_equal(both[:9].reshape(3, 3), oK) and np.array_equal(both[9:].reshape(3, 3), oN)


@pytest.mark.gpu
@pytest.mark.parametrize("method,planar", [(8, False), (4, False), (8, True), (4, True)])
def test_mono_relative_pose_through_the_shim_matches_oracle(oracle, tmp_path, method, planar):
    from test_gpu_parity import _mono_scene
    _build()
    K, x1, x2, R, t = _mono_scene(400, 300 + method + planar, planar=planar, noise=0.15, outliers=0.2)
    thr = 1.0 if method == 8 else 0.1
    inp, outp = tmp_path / "mi.bin", tmp_path / "mo.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<3i", len(x1), method, method))
        f.write(np.ascontiguousarray(K, np.float64).tobytes())
        f.write(struct.pack("<2d", thr, thr))
        f.write(np.ascontiguousarray(x1, np.float32).tobytes()); f.write(np.ascontiguousarray(x2, np.float32).tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_mono_pose"), str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = open(outp, "rb").read()
    ue_in, ue_out, success, nin = struct.unpack("<4i", raw[:16])
    Rt = np.frombuffer(raw[16:16 + 96], np.float64)
    pts = np.frombuffer(raw[16 + 96:], np.float32).reshape(nin, 4)
    op = oracle.mono_params(method=method)
    op.ESSENTIAL_THRESHOLD = thr; op.HOMOGRAPHY_THRESHOLD = thr
    import ctypes as C
    want_ue = bool(oracle.lib().orc_select_estimation_method(x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), len(x1), 10))
    assert bool(ue_in) == want_ue
    ok, ue, Ro, to, in1, in2, mask = oracle.estimate_relative_pose(op, want_ue, x1, x2, K)
    assert (bool(success), bool(ue_out), nin) == (ok, ue, len(in1))
    assert np.array_equal(Rt[:9].view(np.uint64), Ro.ravel().view(np.uint64)) and np.array_equal(Rt[9:].view(np.uint64), to.view(np.uint64))
    assert np.array_equal(pts[:, :2], in1) and np.array_equal(pts[:, 2:], in2)


@pytest.mark.gpu
def test_shim_match_features_hamming_branch(oracle, tmp_path):
    """match_features with FEATURE_DETECTOR = "ORB" (VO_utility.cpp:520-524) through the C++ surface: CV_8U descriptors, Hamming 2-NN +
    ratio, appended after what the vector already holds."""
    _build()
    rng = np.random.default_rng(4)
    n1, n2, nb, ratio = 700, 900, 32, 0.9
    a = rng.integers(0, 256, (n1, nb), dtype=np.uint8); b = rng.integers(0, 256, (n2, nb), dtype=np.uint8)
    b[600:640] = b[20:60]                                # duplicated train rows: those queries fail the ratio test on a tie
    a[:200] = b[100:300]; a[:200, 0] ^= 0x11            # two bits away from one train row each: clear winners
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(struct.pack("<iiif", n1, n2, nb, ratio) + a.tobytes() + b.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_match_binary"), str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = outp.read_bytes()
    m = struct.unpack("<i", raw[:4])[0]
    rec = np.frombuffer(raw[4:], np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")]))
    om = oracle.match_hamming(a, b, ratio)
    assert m == len(om) >= 200
    assert np.array_equal(rec["q"], om["queryIdx"]) and np.array_equal(rec["t"], om["trainIdx"]) and np.array_equal(rec["d"], om["distance"])


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [61, 32])
def test_shim_seven_argument_match_features_applies_l2_to_binary_rows(tmp_path, nb):
    """VO_utility.cpp:551-573: the mono loop's match_features overload constructs BFMatcher(NORM_L2) whatever FEATURE_DETECTOR says; on the
    CV_8U rows of AKAZE (61 bytes) / ORB (32) OpenCV sums squared byte differences in integers and takes the float square root.  Against a
    numpy integer brute force (no oracle): 2 nearest with ties to the lower train index, d0 < ratio * d1 in float."""
    _build()
    rng = np.random.default_rng(6)
    n1, n2, ratio = 500, 700, 0.9
    a = rng.integers(0, 256, (n1, nb), dtype=np.uint8); b = rng.integers(0, 256, (n2, nb), dtype=np.uint8)
    a[:150] = b[100:250]; a[:150, 3] ^= 0x05                                  # near copies: clear winners
    b[600:620] = b[100:120]                                                   # duplicated train rows: ties
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(struct.pack("<iiif", -n1, n2, nb, ratio) + a.tobytes() + b.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_match_binary"), str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = outp.read_bytes()
    m = struct.unpack("<i", raw[:4])[0]
    rec = np.frombuffer(raw[4:], np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")]))
    d2 = ((a[:, None, :].astype(np.int64) - b[None, :, :].astype(np.int64)) ** 2).sum(axis=2)
    order = np.argsort(d2, axis=1, kind="stable")[:, :2]                      # stable: the lower train index first among equals
    want = []
    for q in range(n1):
        f0 = np.sqrt(np.float32(d2[q, order[q, 0]])); f1 = np.sqrt(np.float32(d2[q, order[q, 1]]))
        if f0 < np.float32(ratio) * f1:
            want.append((q, int(order[q, 0]), f0))
    assert m == len(want) >= 120                                              # the 130 near copies whose train row is not duplicated, and a few more
    assert [(int(r["q"]), int(r["t"])) for r in rec] == [(q, t) for q, t, _ in want]
    assert np.array_equal(rec["d"], np.array([d for _, _, d in want], np.float32))


def sift_like_rows(rng, n, base=None, noise=6.0):
    """Rows shaped like cv::SIFT's CV_32F output: 128 non-negative integers (0..255) of norm ~512, stored as floats."""
    a = np.abs(rng.normal(size=(n, 128))) ** 2 if base is None else base + rng.normal(size=base.shape) * noise
    a = np.clip(a, 0, None)
    a = a / np.linalg.norm(a, axis=1, keepdims=True) * 512.0
    return np.clip(np.rint(a), 0, 255).astype(np.float32)


@pytest.mark.gpu
def test_shim_match_features_sift_arm_of_the_l2_branch(tmp_path, oracle):
    """VO_utility.cpp:525-529: FEATURE_DETECTOR == "SIFT" shares BFMatcher(NORM_L2) with "SURF"; the rows are 128 floats whatever
    SURF_EXTENDED says (here it is false, so the context's own rows are 64 wide)."""
    import struct
    import subprocess
    rng = np.random.default_rng(77)
    n1, n2, ratio = 1500, 1700, 0.8
    b = sift_like_rows(rng, n2)
    a = sift_like_rows(rng, n1)
    a[:900] = sift_like_rows(rng, 900, base=b[200:1100].astype(np.float64))       # noisy copies: clear winners
    b[1500:1540] = b[300:340]                                                      # duplicated train rows: ties fail the ratio test
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(struct.pack("<iiif", n1, n2, -128, ratio) + a.tobytes() + b.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_match_binary"), str(inp), str(outp)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = outp.read_bytes()
    m = struct.unpack("<i", raw[:4])[0]
    rec = np.frombuffer(raw[4:], np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")]))
    om = oracle.match(a, b, ratio)
    assert m == len(om) >= 800
    assert np.array_equal(rec["q"], om["queryIdx"]) and np.array_equal(rec["t"], om["trainIdx"])
    assert np.array_equal(rec["d"].view(np.uint32), om["distance"].view(np.uint32))
