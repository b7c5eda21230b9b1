"""detect_features' "SIFT" branch (VO_utility.cpp:107-112: SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute) on the GPU against
the CPU oracle (oracle/o_sift.c), through the C ABI (uvo_sift_detect) and through the C++ uvo_libraries surface.

Bit-exact: keypoints (all seven fields) and the 128 descriptor entries, every pyramid layer.  The oracle is the builder's
restatement of OpenCV 4.5's scalar paths; parity against OpenCV itself is unpinned (o_sift.c header)."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 1920, 1080, 8192)
    yield c
    c.close()


def scene_image(w, h, seed=123, k=0):
    from ergo_uvo_amd import synth
    return synth.stereo_pair(synth.Scene(seed, w), k, w, h)[0]


def same(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def check(ctx, oracle, img, **kw):
    kg, dg = ctx.sift_detect(img, **kw)
    ko, do = oracle.sift_detect(np.asarray(img.cpu() if hasattr(img, "cpu") else img), **kw)
    assert len(kg) == len(ko), (len(kg), len(ko))
    assert same(kg, ko), np.where(kg.view(np.uint8).reshape(len(kg), -1) != ko.view(np.uint8).reshape(len(ko), -1))[0][:5]
    assert same(dg, do), np.where((dg != do).any(axis=1))[0][:5]
    return kg, dg


@pytest.mark.parametrize("w,h", [(203, 131), (97, 64), (640, 360)])
def test_pyramid_layers_bit_exact(ctx, oracle, w, h):
    """createInitialImage + buildGaussianPyramid + buildDoGPyramid: every Gaussian layer of every octave, and the differences."""
    img = scene_image(640, 360)[:h, :w].copy()
    ctx.sift_detect(img)
    n_oct = int(np.rint(np.log2(2 * min(w, h)) - 2)) + 1
    for o in range(n_oct):
        prev = None
        for l in range(6):
            g = ctx.sift_layer(o, l)
            og = oracle.sift_gauss_layer(img, o, l)
            assert same(g, og), (o, l, float(np.max(np.abs(g - og))))
            if prev is not None:
                assert same(ctx.sift_layer(o, l - 1, dog=True), og - prev), (o, l)
            prev = og
    with pytest.raises(Exception):
        ctx.sift_layer(n_oct, 0)


@pytest.mark.parametrize("w,h,seed", [(640, 360, 123), (203, 131, 7), (97, 64, 9), (1280, 720, 11)])
def test_sift_detect_bit_exact(ctx, oracle, w, h, seed):
    img = scene_image(max(w, 320), max(h, 180), seed)[:h, :w].copy()
    kg, dg = check(ctx, oracle, img)
    assert len(kg) > 20
    octave = (kg["octave"] & 255).astype(np.int64); octave[octave >= 128] -= 256
    assert octave.min() == -1 and octave.max() >= 2       # several octaves took part


def test_sift_detect_1080p_and_retain_best(ctx, oracle):
    """The headline geometry: more than 10000 keypoints survive at 1920 x 1080, so retainBest(10000) is active (VOU:109)."""
    img = scene_image(1920, 1080, 20250910)
    kg, dg = check(ctx, oracle, img)
    assert len(kg) >= 10000
    k_all, _ = ctx.sift_detect(img, nfeatures=0, cap=1 << 16)
    assert len(k_all) > len(kg)
    cut = np.sort(k_all["response"])[::-1][9999]
    assert same(kg, k_all[k_all["response"] >= cut])


@pytest.mark.parametrize("kw", [dict(nfeatures=300), dict(n_octave_layers=2), dict(n_octave_layers=5, sigma=1.2), dict(contrast_threshold=0.08, edge_threshold=4.0),
                                dict(nfeatures=0, contrast_threshold=0.01)])
def test_sift_parameters(ctx, oracle, kw):
    img = scene_image(640, 360, 31)
    check(ctx, oracle, img, **kw)


def test_sift_blobs_flat_and_noise(ctx, oracle):
    y, x = np.mgrid[0:120, 0:160]
    for s in (3.0, 8.0):
        img = np.clip(40 + 160 * np.exp(-((x - 70.3) ** 2 + (y - 55.6) ** 2) / (2 * s * s)), 0, 255).round().astype(np.uint8)
        kg, _ = check(ctx, oracle, img)
        assert len(kg) >= 1 and abs(kg["x"][0] - 70.55) < 0.08
    flat = np.full((64, 80), 77, np.uint8)
    kg, dg = check(ctx, oracle, flat)
    assert len(kg) == 0 and dg.shape == (0, 128)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (150, 210), dtype=np.uint8)      # dense extrema, many refinements that wander and are dropped
    check(ctx, oracle, noise, nfeatures=0)
    check(ctx, oracle, noise)


def test_sift_from_device_and_strided_images(ctx, oracle):
    import torch
    img = scene_image(640, 360, 5)
    k0, d0 = check(ctx, oracle, img)
    kd, dd = ctx.sift_detect(torch.from_numpy(img).cuda())
    assert same(kd, k0) and same(dd, d0)
    # a padded host image through the C ABI directly (row pitch > width), and a window of a larger device image
    import ctypes as C
    from ergo_uvo_amd import KP_DTYPE
    pad = np.zeros((360, 704), np.uint8); pad[:, :640] = img
    n = C.c_int(0)
    kps = np.zeros(16384, KP_DTYPE); desc = np.zeros((16384, 128), np.float32)
    ctx._check(ctx._lib.uvo_sift_detect(ctx._h, pad.ctypes.data_as(C.c_void_p), 640, 360, 704, 0, 10000, 3, 0.03, 10.0, 1.6,
                                        kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 16384, C.byref(n)))
    assert same(kps[:n.value], k0) and same(desc[:n.value], d0)
    # keypoints only / count only
    ctx._check(ctx._lib.uvo_sift_detect(ctx._h, pad.ctypes.data_as(C.c_void_p), 640, 360, 704, 0, 10000, 3, 0.03, 10.0, 1.6, None, None, 0, C.byref(n)))
    assert n.value == len(k0)
    # the SURF path of the same context is untouched by the SIFT workspace
    ks, ds = ctx.detect_features(img)
    ko, do = oracle.surf(img, 1500.0)
    assert same(ks, ko) and same(ds, do)


def test_sift_refusals(ctx):
    import ctypes as C
    img = scene_image(640, 360, 5)
    with pytest.raises(Exception, match="capacity"):
        ctx.sift_detect(img, cap=50)
    with pytest.raises(Exception):
        ctx.sift_detect(img, n_octave_layers=0)
    with pytest.raises(Exception):
        ctx.sift_detect(img, sigma=0.4)
    with pytest.raises(Exception):
        ctx.sift_detect(np.zeros((8, 8), np.uint8))
    k, d = ctx.sift_detect(img)                           # still usable afterwards
    assert len(k) > 100


def _shim_detect(tmp_path, img, name, min_hessian=1500):
    from ergo_uvo_amd import KP_DTYPE
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ergo_uvo_amd", "shim"), "-s"])
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    h, w = img.shape
    inp.write_bytes(struct.pack("<iii8s", w, h, min_hessian, name.encode()) + img.tobytes())
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "shim_detect"), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    if res.returncode != 0:
        return res.returncode, res.stderr, None
    raw = outp.read_bytes()
    n, cols = struct.unpack("<ii", raw[:8])
    kps = np.frombuffer(raw[8:8 + 28 * n], KP_DTYPE)
    desc = np.frombuffer(raw[8 + 28 * n:], np.float32).reshape(n, cols)
    return 0, kps, desc


def test_shim_detect_features_sift_and_surf_branches(tmp_path, oracle):
    """detect_features(img, keypoints, descriptors) with FEATURE_DETECTOR = "SIFT" / "SURF" through the C++ surface; "AKAZE" is refused."""
    img = scene_image(640, 360, 17)
    rc, kps, desc = _shim_detect(tmp_path, img, "SIFT")
    assert rc == 0, kps
    ko, do = oracle.sift_detect(img)
    assert desc.shape[1] == 128 and same(kps, ko) and same(desc, do)
    rc, kps, desc = _shim_detect(tmp_path, img, "SURF")
    assert rc == 0 and desc.shape[1] == 64 and len(kps) > 100
    rc, err, _ = _shim_detect(tmp_path, img, "AKAZE")
    assert rc == 4 and "FEATURE_DETECTOR" in err
