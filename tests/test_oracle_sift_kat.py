"""Known-answer and property tests of the SIFT restatement (oracle/o_sift.c), CPU only.

OpenCV is absent and the reference ships no SIFT vectors (VO_utility.cpp:107-112 only calls SIFT::create(10000, 3, 0.03, 10, 1.6)
->detectAndCompute), so what can be pinned here are the facts the algorithm's definition fixes: the numeric helpers against libm,
the filter taps, scale covariance on blobs, the octave packing, the keypoint order / duplicate / retainBest rules, the descriptor's
format.  Parity against OpenCV itself stays unpinned (o_sift.c header)."""
import ctypes as C

import numpy as np
import pytest


def blob(w, h, x0, y0, s, amp=160, base=40):
    y, x = np.mgrid[0:h, 0:w]
    return np.clip(base + amp * np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * s * s)), 0, 255).round().astype(np.uint8)


def kp_sorted(k):
    """KeyPoint_LessThan (features2d keypoint.cpp): x, y ascending, size descending, angle ascending, response / octave descending."""
    key = np.lexsort((-k["octave"].astype(np.int64), -k["response"].astype(np.float64), k["angle"], -k["size"].astype(np.float64), k["y"], k["x"]))
    return np.array_equal(key, np.arange(len(k)))


def test_exp32f_and_exp2_against_libm(oracle):
    lib = oracle.lib()
    lib.orc_exp32f.restype = C.c_float; lib.orc_exp32f.argtypes = [C.c_float]
    lib.orc_exp2f_det.restype = C.c_float; lib.orc_exp2f_det.argtypes = [C.c_float]
    xs = np.linspace(-30, 8, 1501, dtype=np.float32)
    e = np.array([lib.orc_exp32f(float(x)) for x in xs], np.float64)
    r = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(e - r) / r) < 2.5e-7            # hal::exp32f: table + degree-4 polynomial, a few float ulps
    assert lib.orc_exp32f(0.0) == 1.0
    for x in (-1.25, -0.5, 0.0, 1.0 / 3, 0.5, 1.0, 2.75):
        assert lib.orc_exp2f_det(float(np.float32(x))) == np.float32(2.0 ** float(np.float32(x)))


def test_gaussian_taps(oracle):
    lib = oracle.lib()
    lib.orc_sift_gauss_kernel.argtypes = [C.c_double, C.c_void_p]
    k = (C.c_float * 64)()
    for sigma in (1.2489996, 1.2262735, 1.5450078, 1.9465878, 2.452547, 3.09, 3.2):
        n = lib.orc_sift_gauss_kernel(sigma, k)
        a = np.array(k[:n], np.float64)
        assert n == (int(np.rint(sigma * 8 + 1)) | 1) and n % 2 == 1          # GaussianBlur's automatic size for CV_32F
        assert abs(a.sum() - 1) < 1e-6 and np.array_equal(a, a[::-1]) and np.all(np.diff(a[: n // 2 + 1]) > 0)
        x = np.arange(n) - (n - 1) / 2
        g = np.exp(-x * x / (2 * sigma * sigma)); g /= g.sum()
        assert np.max(np.abs(a - g)) < 1e-7
    sig = (C.c_double * 8)()
    lib.orc_sift_layer_sigmas.argtypes = [C.c_int, C.c_double, C.c_void_p]
    lib.orc_sift_layer_sigmas(3, 1.6, sig)
    k3 = 2 ** (1 / 3)
    tot = 1.6
    for i in range(1, 6):                                 # each blur takes layer i-1's total sigma to k times it
        assert abs(np.hypot(tot, sig[i]) - tot * k3) < 1e-12
        tot *= k3
    lib.orc_sift_num_octaves.argtypes = [C.c_int, C.c_int]
    assert lib.orc_sift_num_octaves(1920, 1080) == int(np.rint(np.log2(2160) - 2)) + 1 == 10
    assert lib.orc_sift_num_octaves(640, 360) == 8


def test_gaussian_pyramid_layers(oracle):
    img = blob(96, 64, 40.0, 30.0, 4.0)
    g0 = oracle.sift_gauss_layer(img, 0, 0)
    assert g0.shape == (128, 192)                         # firstOctave = -1: the doubled image
    assert abs(g0.mean() - img.mean()) < 0.5 and g0.max() <= 255.0 and g0.min() >= 0
    g3 = oracle.sift_gauss_layer(img, 0, 3)
    g10 = oracle.sift_gauss_layer(img, 1, 0)
    assert g10.shape == (64, 96) and np.array_equal(g10, g3[::2, ::2])     # INTER_NEAREST halving of layer nOctaveLayers
    flat = np.full((40, 56), 93, np.uint8)
    for o, l in ((0, 0), (0, 5), (2, 4)):
        g = oracle.sift_gauss_layer(flat, o, l)
        assert np.max(np.abs(g - 93)) < 1e-3              # taps sum to one; borders are reflected
    with pytest.raises(ValueError):
        oracle.sift_gauss_layer(img, 0, 6)


@pytest.mark.parametrize("s", [3.0, 5.0, 8.0, 11.0])
def test_blob_is_found_at_its_place_and_scale(oracle, s):
    x0, y0 = 70.3, 55.6
    kp, d = oracle.sift_detect(blob(160, 120, x0, y0, s))
    assert len(kp) >= 1
    # one extremum (with several orientation peaks: the blob is isotropic); 4.5's doubled base image is sampled half a pixel off,
    # which puts every keypoint a quarter of a pixel down-right of the true place
    assert np.all(kp["x"] == kp["x"][0]) and np.all(kp["y"] == kp["y"][0]) and np.all(kp["size"] == kp["size"][0])
    assert abs(kp["x"][0] - (x0 + 0.25)) < 0.08 and abs(kp["y"][0] - (y0 + 0.25)) < 0.08
    assert abs(kp["size"][0] / s - 1.775) < 0.03          # scale covariance: diameter / blob sigma is constant
    assert np.all(np.diff(kp["angle"]) > 0) and np.all((kp["angle"] >= 0) & (kp["angle"] < 360))
    assert np.all(kp["response"] > 0.03 / 3) and np.all(kp["class_id"] == -1)


def test_octave_packing_order_and_descriptor_format(oracle, scene_small):
    img = scene_small[0][0]
    kp, d = oracle.sift_detect(img)
    assert 800 < len(kp) < 10000 and d.shape == (len(kp), 128)
    octave = (kp["octave"] & 255).astype(np.int64); octave[octave >= 128] -= 256
    layer = (kp["octave"] >> 8) & 255
    xi = ((kp["octave"] >> 16) & 255) / 255.0 - 0.5
    assert octave.min() == -1 and octave.max() <= 6 and layer.min() >= 1 and layer.max() <= 3 and np.all(np.abs(xi) <= 0.5 + 1 / 255)
    size = 1.6 * 2.0 ** ((layer + xi) / 3.0) * 2.0 ** octave * 2       # kpt.size of adjustLocalExtrema, after the halving
    assert np.max(np.abs(size / kp["size"] - 1)) < 2e-3                 # (xi is stored in 8 bits)
    assert kp_sorted(kp)
    dup = (np.diff(kp["x"]) == 0) & (np.diff(kp["y"]) == 0) & (np.diff(kp["size"]) == 0) & (np.diff(kp["angle"]) == 0)
    assert not dup.any()
    assert np.all(d == np.rint(d)) and d.min() >= 0 and d.max() <= 255
    nrm = np.linalg.norm(d.astype(np.float64), axis=1)
    assert np.all(np.abs(nrm - 512) < 8)                  # 512 / |clipped row|, each entry rounded
    # SIFT_IMG_BORDER = 5 pixels of the doubled image
    assert 2 <= kp["x"].min() and 2 <= kp["y"].min() and kp["x"].max() < img.shape[1] - 2 and kp["y"].max() < img.shape[0] - 2
    kp2, d2 = oracle.sift_detect(img)
    assert np.array_equal(kp.view(np.uint8), kp2.view(np.uint8)) and np.array_equal(d, d2)
    kp3, _ = oracle.sift_detect(img, descriptors=False)
    assert np.array_equal(kp.view(np.uint8), kp3.view(np.uint8))


def test_retain_best_and_thresholds(oracle, scene_small):
    img = scene_small[0][0]
    kp, d = oracle.sift_detect(img)
    n = 300
    kpn, dn = oracle.sift_detect(img, nfeatures=n)
    cut = np.sort(kp["response"])[::-1][n - 1]
    keep = kp["response"] >= cut                          # retainBest keeps ties at the cut
    assert len(kpn) == keep.sum() >= n
    assert np.array_equal(kpn.view(np.uint8), kp[keep].view(np.uint8)) and np.array_equal(dn, d[keep])
    kp_all, _ = oracle.sift_detect(img, nfeatures=0, descriptors=False)
    assert len(kp_all) == len(kp)                         # fewer than 10000 here: the cut was not active
    hi, _ = oracle.sift_detect(img, contrast_threshold=0.08, descriptors=False)
    assert 0 < len(hi) < len(kp) and hi["response"].min() >= 0.08 / 3 - 1e-7
    # a stricter edge threshold (smaller r) only removes keypoints
    ed, _ = oracle.sift_detect(img, edge_threshold=3.0, descriptors=False)
    assert 0 < len(ed) < len(kp)
    s = set(map(bytes, kp.view(np.uint8).reshape(len(kp), -1)))
    assert all(bytes(r) in s for r in ed.view(np.uint8).reshape(len(ed), -1))


def test_descriptor_follows_the_image_under_a_quarter_turn(oracle, scene_small):
    """np.rot90 maps pixel (x, y) to (y, w - 1 - x) exactly.  The pyramid is not bit-identical under it (rows are filtered before
    columns, the doubled image is sampled off-centre), so keypoints are paired by position and size and the rest compared loosely: a
    rotated image gives rotated keypoints (gradient (gx, gy) -> (gy, -gx): the angle drops by 90) whose descriptors, taken relative
    to their own orientation, stay close."""
    img = np.ascontiguousarray(scene_small[0][0][60:260, 200:440])
    kp, d = oracle.sift_detect(img)
    kr, dr = oracle.sift_detect(np.ascontiguousarray(np.rot90(img)))
    assert len(kp) > 100 and abs(len(kr) - len(kp)) <= 0.2 * len(kp)
    w = img.shape[1]
    hits = 0
    strong = np.where(kp["response"] > 0.03)[0]           # weak extrema come and go with the half-pixel asymmetry
    for i in strong:
        # expected place in the rotated image; the quarter-pixel bias does not rotate with the image
        ex, ey = kp["y"][i], (w - 1) - kp["x"][i] + 0.5
        dd = np.hypot(kr["x"] - ex, kr["y"] - ey) + np.abs(kr["size"] - kp["size"][i])
        da = np.abs(((kr["angle"] - (kp["angle"][i] - 90)) + 180) % 360 - 180)
        j = np.where((dd < 0.6) & (da < 6))[0]
        if len(j):
            hits += 1
            assert np.linalg.norm(dr[j[0]] - d[i]) < 0.35 * 512
    assert len(strong) > 40 and hits >= 0.6 * len(strong)


def test_pyramid_against_scipy(oracle, scene_small):
    """An independent statement of the pyramid: scipy.ndimage.correlate1d with mode='mirror' (= BORDER_REFLECT_101) and the same taps,
    in float64, layer after layer as buildGaussianPyramid chains them.  The oracle accumulates in float (in OpenCV's order), so the
    comparison is to 2e-3 on a 0..255 scale, far below anything a wrong border, tap count, sigma or chaining would leave."""
    ndi = pytest.importorskip("scipy.ndimage")
    img = np.ascontiguousarray(scene_small[0][0][40:168, 100:292])
    lib = oracle.lib()
    lib.orc_sift_gauss_kernel.argtypes = [C.c_double, C.c_void_p]
    lib.orc_sift_layer_sigmas.argtypes = [C.c_int, C.c_double, C.c_void_p]

    def blur(a, sigma):
        k = (C.c_float * 64)()
        n = lib.orc_sift_gauss_kernel(float(sigma), k)
        taps = np.array(k[:n], np.float64)
        return ndi.correlate1d(ndi.correlate1d(a, taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")

    # createInitialImage: bilinear doubling with OpenCV's half-pixel convention, then the blur that brings sigma 1.0 to 1.6
    h, w = img.shape
    def lin(n_dst, n_src):
        f = (np.arange(n_dst) + 0.5) * 0.5 - 0.5
        s = np.floor(f).astype(int); f = f - s
        f[s < 0] = 0; s[s < 0] = 0
        f[s + 1 >= n_src] = 0; s[s + 1 >= n_src] = n_src - 1
        return s, np.minimum(s + 1, n_src - 1), f
    sx, sx1, fx = lin(2 * w, w); sy, sy1, fy = lin(2 * h, h)
    a = img.astype(np.float64)
    hor = a[:, sx] * (1 - fx) + a[:, sx1] * fx
    dbl = hor[sy] * (1 - fy)[:, None] + hor[sy1] * fy[:, None]
    sig = (C.c_double * 8)()
    lib.orc_sift_layer_sigmas(3, 1.6, sig)
    g = blur(dbl, np.sqrt(1.6 ** 2 - 4 * 0.5 ** 2))
    for o in range(3):
        layers = [g]
        for i in range(1, 6):
            layers.append(blur(layers[-1], sig[i]))
        for i, ref in enumerate(layers):
            got = oracle.sift_gauss_layer(img, o, i)
            assert got.shape == ref.shape
            assert np.max(np.abs(got - ref)) < 2e-3, (o, i, float(np.max(np.abs(got - ref))))
        g = layers[3][::2, ::2]                           # the next octave starts from layer nOctaveLayers, every second pixel


def test_extrema_against_a_numpy_statement(oracle):
    """Every keypoint's (octave, layer, row, column) must be a 26-neighbour extremum of the DoG stack above the contrast pre-threshold
    -- or reachable from one by adjustLocalExtrema's at most five unit steps -- and every strong, well-conditioned extremum must be
    represented: the stack is rebuilt from the oracle's own Gaussian layers, the extremum test is scipy's maximum / minimum filter."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:96, 0:128]
    img = np.full((96, 128), 70.0)
    for _ in range(25):
        s = rng.uniform(1.5, 5.0)
        img += rng.uniform(50, 120) * np.exp(-((x - rng.uniform(12, 116)) ** 2 + (y - rng.uniform(12, 84)) ** 2) / (2 * s * s))
    img = np.clip(img, 0, 255).astype(np.uint8)
    kp, _ = oracle.sift_detect(img, descriptors=False)
    assert len(kp) > 15
    octv = (kp["octave"] & 255).astype(np.int64); octv[octv >= 128] -= 256
    layer = (kp["octave"] >> 8) & 255
    stacks = {}
    for o in sorted(set((octv + 1).tolist())):
        g = np.stack([oracle.sift_gauss_layer(img, int(o), l) for l in range(6)])
        d = g[1:] - g[:-1]                                 # float32 differences, as buildDoGPyramid
        mx = ndi.maximum_filter(d, size=3, mode="nearest"); mn = ndi.minimum_filter(d, size=3, mode="nearest")
        ext = (np.abs(d) > 1) & (((d > 0) & (d >= mx)) | ((d < 0) & (d <= mn)))       # threshold = floor(0.5 * 0.03 / 3 * 255) = 1
        ext[0] = ext[-1] = False
        ext[:, :5] = ext[:, -5:] = False; ext[:, :, :5] = ext[:, :, -5:] = False      # SIFT_IMG_BORDER
        stacks[int(o)] = ext
    for i in range(len(kp)):
        o = int(octv[i] + 1)
        scale = 2.0 ** octv[i]
        c, r, l = kp["x"][i] / scale, kp["y"][i] / scale, int(layer[i])
        ext = stacks[o]
        l0, l1 = max(l - 5, 1), min(l + 5, 3)
        r0, r1 = int(max(np.floor(r) - 6, 0)), int(np.ceil(r) + 7)
        c0, c1 = int(max(np.floor(c) - 6, 0)), int(np.ceil(c) + 7)
        assert ext[l0:l1 + 1, r0:r1, c0:c1].any(), (i, kp[i])
