"""get_image (VO_utility.cpp:337-379): resize INTER_AREA -> RGB2GRAY -> undistort -> CLAHE.

CPU: known-answer tests of the oracle restatement (oracle/o_preproc.c) -- closed forms that do not depend on the
restatement itself.  GPU: the HIP path (ergo_uvo_amd/csrc/preproc.hip, through the C ABI uvo_get_image) must equal
the oracle byte for byte, stage by stage and end to end."""
import numpy as np
import pytest


def _rgb(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (128 + 80 * np.sin(xx / 9.0) * np.cos(yy / 7.0))[..., None] + rng.normal(0, 12, (h, w, 3))
    base += np.array([10, -5, 20])
    return np.clip(base, 0, 255).astype(np.uint8)


def _cam(w, h, s=1.0):
    K = np.array([[0.9 * w * s, 0, 0.51 * w * s], [0, 0.92 * w * s, 0.49 * h * s], [0, 0, 1.0]])
    newK = np.array([[0.84 * w * s, 0, 0.50 * w * s], [0, 0.86 * w * s, 0.50 * h * s], [0, 0, 1.0]])
    return K, np.array([-0.21, 0.06, 0.0012, -0.0017]), newK


# ------------------------------------------------------------------ oracle KATs (CPU)
def test_rgb2gray_fixed_point(oracle):
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [12, 200, 99]]], np.uint8)
    g = oracle.rgb2gray(px)[0]
    want = [255, 0, (255 * 9798 + 16384) >> 15, (255 * 19235 + 16384) >> 15, (255 * 3735 + 16384) >> 15,
            (12 * 9798 + 200 * 19235 + 99 * 3735 + 16384) >> 15]
    assert g.tolist() == want
    assert 9798 + 19235 + 3735 == 1 << 15                     # white stays white


def test_resize_area_c3_integer_scales(oracle):
    img = _rgb(48, 96, 3)
    r2 = oracle.resize_area_c3(img, 48, 24)                       # 2x2 blocks: (sum + 2) >> 2
    blocks = img.reshape(24, 2, 48, 2, 3).astype(np.int64).sum(axis=(1, 3))
    assert np.array_equal(r2, ((blocks + 2) >> 2).astype(np.uint8))
    r3 = oracle.resize_area_c3(img, 32, 16)                       # 3x3 blocks: round-half-even of sum * (1.f/9)
    b3 = img.reshape(16, 3, 32, 3, 3).astype(np.int64).sum(axis=(1, 3))
    want = np.rint((b3.astype(np.float32) * np.float32(1.0 / 9)).astype(np.float32)).astype(np.uint8)
    assert np.array_equal(r3, want)


def test_resize_area_c3_fractional_constant_and_mean(oracle):
    flat = np.full((45, 80, 3), (37, 150, 201), np.uint8)
    assert np.array_equal(oracle.resize_area_c3(flat, 64, 36), np.broadcast_to(flat[0, 0], (36, 64, 3)))
    img = _rgb(90, 160, 4)
    r = oracle.resize_area_c3(img, 100, 56).astype(np.float64)    # area averaging preserves the mean (up to rounding)
    assert abs(r.mean() - img.mean()) < 0.6
    with pytest.raises(ValueError):
        oracle.resize_area_c3(img, 200, 112)


def test_undistort_identity_and_shift(oracle):
    g = oracle.rgb2gray(_rgb(60, 100, 5))
    K = np.array([[80.0, 0, 50], [0, 82.0, 30], [0, 0, 1]])
    assert np.array_equal(oracle.undistort(g, K, [0, 0, 0, 0], K), g)                  # same camera, no distortion: identity
    newK = K.copy(); newK[0, 2] += 3; newK[1, 2] -= 2                                  # principal point moved by whole pixels
    u = oracle.undistort(g, K, [0, 0, 0, 0], newK)
    assert np.array_equal(u[:-2, 3:], g[2:, :-3])                                      # pure integer shift ...
    assert not u[:, :3].any() and not u[-2:, :].any()                                  # ... with BORDER_CONSTANT 0 where the source is outside
    half = K.copy(); half[0, 2] += 0.5                                                 # half-pixel shift: exact bilinear average, round half up
    uh = oracle.undistort(g, K, [0, 0, 0, 0], half)
    want = (g[:, :-1].astype(int) * 16384 + g[:, 1:].astype(int) * 16384 + 16384) >> 15
    assert np.array_equal(uh[:, 1:], want.astype(np.uint8))


def test_clahe_properties(oracle):
    flat = np.full((64, 96), 77, np.uint8)
    out = oracle.clahe(flat, 3.0)
    assert len(np.unique(out)) == 1                                                    # a constant image stays constant
    g = oracle.rgb2gray(_rgb(64, 96, 6))
    big = oracle.clahe(g, 1e9)                                                         # no clipping: plain tile equalisation, monotone per tile
    centre = (slice(4, 5), slice(6, 7))                                                # the centre pixel of tile (0, 0) blends that tile's LUT alone
    tile = g[:8, :12]
    cdf = np.cumsum(np.bincount(tile.ravel(), minlength=256))
    lut = np.rint((cdf * np.float32(255.0 / tile.size)).astype(np.float32)).clip(0, 255).astype(np.uint8)
    # pixel (y=3.5, x=5.5) does not exist; check the two LUT-pure rows/cols instead: tyf = y/8 - .5, txf = x/12 - .5
    y, x = 4, 6                                                                        # tyf = 0.0, txf = 0.0 -> weights (1, 0): tile (0,0) only
    assert big[y, x] == lut[g[y, x]]
    odd = oracle.clahe(oracle.rgb2gray(_rgb(61, 93, 7)), 3.0)                          # not divisible by 8: reflect-101 extension
    assert odd.shape == (61, 93)


def test_get_image_composition(oracle):
    img = _rgb(96, 160, 8)
    K, d, newK = _cam(80, 48)
    got = oracle.get_image(img, 80, K, d, newK, True, 3)
    step = oracle.clahe(oracle.undistort(oracle.rgb2gray(oracle.resize_area_c3(img, 80, 48)), K, d, newK), 3.0)
    assert np.array_equal(got, step)
    same = oracle.get_image(img, 160, *_cam(160, 96), False, 3)                        # size already right: no resize, no CLAHE
    assert np.array_equal(same, oracle.undistort(oracle.rgb2gray(img), *_cam(160, 96)))


# ------------------------------------------------------------------ HIP parity (GPU)
@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()               # torch's bundled HIP runtime must come up before libuvo_hip.so brings in /opt/rocm's
    import ergo_uvo_amd as uvo
    c = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,dw,clahe,clip", [(96, 160, 80, True, 3), (96, 160, 160, True, 8), (90, 150, 100, True, 3),
                                                (120, 200, 67, False, 3), (135, 243, 81, True, 40), (1080, 1920, 640, True, 3), (360, 640, 320, True, 3),
                                                (720, 1280, 1280, True, 8)])
def test_get_image_bit_exact(ctx, oracle, h, w, dw, clahe, clip):
    img = _rgb(h, w, 11 + h)
    dh = int(h / (w / dw))
    K, d, newK = _cam(dw, dh)
    want = oracle.get_image(img, dw, K, d, newK, clahe, clip)
    got = ctx.get_image(img, dw, K, d, newK, clahe, clip)
    assert got.shape == want.shape
    assert np.array_equal(got, want), (np.argwhere(got != want)[:5], int((got != want).sum()))
    again = ctx.get_image(img, dw, K, d, newK, clahe, clip)                            # cached maps
    assert np.array_equal(again, want)
    K2 = K.copy(); K2[0, 2] += 1.25                                                    # new camera: maps must be rebuilt
    assert np.array_equal(ctx.get_image(img, dw, K2, d, newK, clahe, clip), oracle.get_image(img, dw, K2, d, newK, clahe, clip))


@pytest.mark.gpu
def test_get_image_device_in_out_feeds_the_detector(ctx, oracle):
    import torch
    import ergo_uvo_amd as uvo
    img = _rgb(360, 640, 21)
    K, d, newK = _cam(320, 180)
    want = oracle.get_image(img, 320, K, d, newK, True, 3)
    timg = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()                       # the library works on its own streams: device inputs must be complete
    dev = ctx.get_image(timg, 320, K, d, newK, True, 3, device_out=True)
    assert np.array_equal(dev.cpu().numpy(), want)
    ctx.set_params(uvo.Params.stereo(SURF_MIN_HESSIAN=300))
    kps, desc = ctx.detect_features(dev)
    okps, odesc = oracle.surf(want, 300)
    assert len(kps) == len(okps) > 20
    assert np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))
    with pytest.raises(uvo.UvoError):
        ctx.get_image(img, 1280, K, d, newK)                                           # enlarging is not provided


# ---- resize_camera_matrix (VO_utility.cpp:658-675): host arithmetic, no GPU needed ----
def test_resize_camera_matrix_zero_distortion_is_the_scaled_matrix():
    from oracle import pyoracle as po
    K = np.array([[1400.0, 0.7, 955.0], [0, 1395.0, 542.0], [0, 0, 1]])
    Ks, newK, dh = po.resize_camera_matrix(1920, 1080, 960, K, [0, 0, 0, 0])
    assert dh == 540
    expect = K / 2.0
    expect[0, 1] = 0.7; expect[2, 2] = 1.0                       # skew kept, K[2][2] restored
    assert np.array_equal(Ks, expect)
    # without distortion the undistorted 9 x 9 grid is the pixel grid: the inscribed rectangle is the whole image and the
    # optimal matrix reproduces fx, fy, cx, cy (the skew is not part of the undistortion model, so cx is only as exact as it)
    K0 = K.copy(); K0[0, 1] = 0
    Ks0, newK0, _ = po.resize_camera_matrix(1920, 1080, 960, K0, [0, 0, 0, 0])
    assert np.allclose(newK0, Ks0, rtol=0, atol=1e-9)


def test_resize_camera_matrix_centred_radial_distortion_stays_centred():
    from oracle import pyoracle as po
    W, H, DW = 1280, 720, 640
    K = np.array([[800.0, 0, (W - 1) / 2.0], [0, 800.0, (H - 1) / 2.0], [0, 0, 1]])
    for k1 in (-0.25, 0.12):
        Ks, newK, dh = po.resize_camera_matrix(W, H, DW, K, [k1, 0.03, 0, 0])
        assert dh == 360
        assert np.isfinite(newK).all() and newK[0, 0] > 0 and newK[1, 1] > 0
        # distortion symmetric about the principal point: the inscribed rectangle stays centred on it
        assert abs(newK[0, 2] - Ks[0, 2]) < 0.5 and abs(newK[1, 2] - Ks[1, 2]) < 0.5
        # barrel (k1 < 0): undistorting stretches the border outwards, so the rectangle inscribed in it is larger in
        # normalised units and alpha = 0 gives a shorter focal length; pincushion the other way round
        assert (newK[0, 0] < Ks[0, 0]) == (k1 < 0) and (newK[1, 1] < Ks[1, 1]) == (k1 < 0)
    # the library's host implementation is the same arithmetic (loads without a GPU: no device call is made)
    import ergo_uvo_amd as uvo
    for d in ([-0.25, 0.03, 1e-3, -2e-3], [0.12, -0.01, 0, 0], [0, 0, 0, 0]):
        a = po.resize_camera_matrix(W, H, DW, K, d)
        b = uvo.resize_camera_matrix(W, H, DW, K, d)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
