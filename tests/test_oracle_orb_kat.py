"""Known-answer tests of the ORB restatement (oracle/o_orb.c; detect_features' ORB branch, VO_utility.cpp:100-105:
ORB::create(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, 10)->detectAndCompute) against independent numpy statements of its parts --
PARITY vs OpenCV is UNPINNED; these pin the restatement to the published method (Rublee et al., ICCV 2011; FAST: Rosten, Drummond):

  * the level geometry: sizes round(size / 1.2^l), the border 32, the features wanted per level a geometric series summing to nfeatures;
  * INTER_LINEAR_EXACT: identity at equal size, constants stay constant, exact 2:1 averaging, and every pixel within the rounding of
    the float bilinear formula at half-pixel centres;
  * FAST-9/16: the score is the largest threshold at which 9 contiguous circle pixels are all darker or all brighter (brute force over
    thresholds), zero elsewhere; the detector keeps strict 3 x 3 maxima in row-major order;
  * retainBest keeps exactly the responses >= the n-th largest (ties included), as a permutation, and is the identity below n;
  * the Harris response is ((sum Ix^2)(sum Iy^2) - (sum IxIy)^2 - 0.04 (sum Ix^2 + sum Iy^2)^2) / (4 * 7 * 255)^4 with Sobel gradients over 7 x 7;
  * the intensity-centroid angle is atan2(m01, m10) over the disc of radius 15 (to fastAtan2's 0.3 degrees);
  * the 7-tap blur is the integer kernel [18 34 49 55 49 34 18] / 2^8 per axis, reflect-101, rounded once;
  * a descriptor bit is I(p0) < I(p1) at the pattern rotated by the angle; rotating image and keypoint by 90 degrees leaves it unchanged;
  * the whole detector: keypoints lie >= 31 pixels inside their level, carry size 31 x 1.2^l, per-level counts within the shares, and the
    descriptors of a shifted image match."""
import numpy as np
import pytest

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _img(w, h, seed):
    from ergo_uvo_amd import synth
    return synth.stereo_pair(synth.Scene(seed, w), 0, w, h)[0]


def test_level_geometry(oracle):
    border, lv, sc = oracle.orb_levels(1920, 1080)
    assert border == 32
    assert np.allclose(sc, 1.2 ** np.arange(8), rtol=1e-6)
    assert [tuple(r[:2]) for r in lv] == [(int(np.rint(1920 / 1.2 ** l)), int(np.rint(1080 / 1.2 ** l))) for l in range(8)]
    assert int(lv[:, 2].sum()) == 10000 and np.all(np.diff(lv[:, 2]) < 0)
    f = 1 / 1.2
    assert np.all(np.abs(lv[:, 2] - 10000 * (1 - f) / (1 - f ** 8) * f ** np.arange(8)) <= 1.5)
    assert oracle.orb_umax().tolist()[:16] == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]      # the disc's row ends (orb.cpp)


def test_linear_exact_resize(oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear_exact(a, 53, 37), a)
    assert np.all(oracle.resize_linear_exact(np.full((40, 60), 173, np.uint8), 50, 33) == 173)
    b = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    half = oracle.resize_linear_exact(b, 30, 20).astype(int)
    blk = b.astype(int).reshape(20, 2, 30, 2).sum(axis=(1, 3))
    assert np.all(np.abs(half - blk / 4.0) <= 0.5 + 1e-9)                                                      # 2:1: the mean of the 2 x 2 block, rounded once
    for (dw, dh) in ((50, 33), (44, 31), (59, 39)):
        out = oracle.resize_linear_exact(b, dw, dh).astype(float)
        fx = np.clip((np.arange(dw) + 0.5) * 60 / dw - 0.5, 0, 59); fy = np.clip((np.arange(dh) + 0.5) * 40 / dh - 0.5, 0, 39)
        x0 = np.minimum(np.floor(fx).astype(int), 58); y0 = np.minimum(np.floor(fy).astype(int), 38)
        ax = (fx - x0)[None, :]; ay = (fy - y0)[:, None]
        bf = b.astype(float)
        ref = (bf[y0][:, x0] * (1 - ax) + bf[y0][:, x0 + 1] * ax) * (1 - ay) + (bf[y0 + 1][:, x0] * (1 - ax) + bf[y0 + 1][:, x0 + 1] * ax) * ay
        assert np.max(np.abs(out - ref)) <= 0.5 + 255 * 2.2 / 256                                               # weights quantised to 1/256 on each axis


def _fast_brute(img, x, y):
    """largest t in 0..254 at which (x, y) has 9 contiguous circle pixels all < v - t or all > v + t; -1 if none"""
    v = int(img[y, x]); ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    best = -1
    for t in range(0, 255):
        ok = False
        for s in range(16):
            arc = [ring[(s + k) % 16] for k in range(9)]
            if all(p < v - t for p in arc) or all(p > v + t for p in arc):
                ok = True; break
        if ok:
            best = t
        else:
            break
    return best


def test_fast_scores_and_maxima(oracle):
    rng = np.random.default_rng(9)
    img = np.clip(rng.normal(128, 40, (48, 64)), 0, 255).astype(np.uint8)
    img[20:30, 30:44] = 230                                                                                      # a bright rectangle: corners
    for thr in (10, 20, 40):
        sc = oracle.fast_scores(img, thr)
        assert not sc[:3].any() and not sc[-3:].any() and not sc[:, :3].any() and not sc[:, -3:].any()
        for y in range(3, 45):
            for x in range(3, 61):
                b = _fast_brute(img, x, y)
                assert int(sc[y, x]) == (b if b >= thr else 0), (thr, x, y, b, sc[y, x])
        k = oracle.fast_detect(img, thr)
        p = np.pad(sc.astype(int), 1)
        want = [(x, y) for y in range(48) for x in range(64) if sc[y, x] and all(sc[y, x] > p[y + 1 + dy, x + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0))]
        assert [(int(a), int(b)) for a, b in zip(k["x"], k["y"])] == want                                        # row-major
        assert np.array_equal(k["response"], np.array([sc[y, x] for x, y in want], np.float32)) and np.all(k["size"] == 7) and np.all(k["angle"] == -1)


def test_retain_best_is_a_threshold_on_the_nth_response(oracle):
    rng = np.random.default_rng(3)
    for n, keep, levels in ((1000, 100, 40), (5000, 777, 15), (50, 49, 5), (64, 1, 3), (4000, 2000, 100000)):
        r = rng.integers(0, levels, n).astype(np.float32)
        perm = oracle.retain_best(r, keep)
        nth = np.sort(r)[::-1][keep - 1]
        assert len(set(perm.tolist())) == len(perm)
        assert sorted(perm.tolist()) == np.flatnonzero(r >= nth).tolist()                                       # every tie of the boundary response stays
        assert np.all(r[perm[:keep - 1]] >= nth) and r[perm[keep - 1]] == nth                                   # nth_element's postcondition
    r = rng.random(30).astype(np.float32)
    assert oracle.retain_best(r, 30).tolist() == list(range(30)) and oracle.retain_best(r, 31).tolist() == list(range(30))
    assert len(oracle.retain_best(r, 0)) == 0


def test_harris_response(oracle):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    f = img.astype(np.int64)
    for (x, y) in ((10, 12), (20, 20), (30, 9)):
        a = b = c = 0
        for i in range(-3, 4):
            for j in range(-3, 4):
                p = lambda dy, dx: f[y + i + dy, x + j + dx]
                ix = (p(0, 1) - p(0, -1)) * 2 + (p(-1, 1) - p(-1, -1)) + (p(1, 1) - p(1, -1))
                iy = (p(1, 0) - p(-1, 0)) * 2 + (p(1, -1) - p(-1, -1)) + (p(1, 1) - p(-1, 1))
                a += ix * ix; b += iy * iy; c += ix * iy
        want = (float(a) * b - float(c) * c - 0.04 * float(a + b) ** 2) / (4 * 7 * 255.0) ** 4
        assert abs(oracle.orb_harris(img, x, y) - want) <= 1e-5 * abs(want) + 1e-12


def test_intensity_centroid_angle(oracle):
    rng = np.random.default_rng(13)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    yy, xx = np.mgrid[-15:16, -15:16]
    umax = oracle.orb_umax()
    disc = np.abs(xx) <= umax[np.abs(yy)]
    assert disc.sum() == 2 * sum(2 * int(umax[v]) + 1 for v in range(1, 16)) + 31
    for (x, y) in ((32, 32), (20, 40), (45, 17)):
        pch = img[y - 15:y + 16, x - 15:x + 16].astype(np.int64)
        m10, m01 = int((xx * pch)[disc].sum()), int((yy * pch)[disc].sum())
        want = np.degrees(np.arctan2(m01, m10)) % 360
        d = abs(oracle.orb_ic_angle(img, x, y) - want)
        assert min(d, 360 - d) <= 0.3
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 3, (64, 1))                                                  # brighter to the right: angle 0; transposed: 90
    assert oracle.orb_ic_angle(ramp, 32, 32) == 0.0 and abs(oracle.orb_ic_angle(ramp.T.copy(), 32, 32) - 90.0) <= 1e-3


def test_blur_is_the_integer_kernel(oracle):
    k = oracle.orb_blur_kernel()
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18]
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0); g /= g.sum()
    assert np.all(np.abs(k - 256 * g) <= 0.5)
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (30, 41), dtype=np.uint8)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")                                                         # numpy "reflect" = BORDER_REFLECT_101
    rows = sum(k[t] * p[:, t:t + 41] for t in range(7))
    out = sum(k[t] * rows[t:t + 30] for t in range(7))
    assert np.array_equal(oracle.orb_blur(img), np.clip((out + (1 << 15)) >> 16, 0, 255).astype(np.uint8))


def test_descriptor_bits_and_rotation(oracle):
    rng = np.random.default_rng(19)
    img = rng.integers(0, 256, (80, 80), dtype=np.uint8)
    pat = oracle.orb_random_pattern()
    assert pat.shape == (512, 2) and pat.min() >= -15 and pat.max() <= 15
    d0 = oracle.orb_describe(img, 40, 40, 0.0, pat)
    bits = np.unpackbits(d0, bitorder="little")
    want = [int(img[40 + pat[2 * i][1], 40 + pat[2 * i][0]] < img[40 + pat[2 * i + 1][1], 40 + pat[2 * i + 1][0]]) for i in range(256)]
    assert bits.tolist() == want
    # the image turned by 90 degrees (x, y) -> (-y, x) about the centre and the angle by +90: the same bits
    rot = np.rot90(img, k=-1).copy()                                                                             # rot[y', x'] with (x', y') = (79 - y, x)
    d90 = oracle.orb_describe(rot, 79 - 40, 40, 90.0, pat)
    assert np.array_equal(d90, d0)
    a = 37.0
    d = oracle.orb_describe(img, 40, 40, a, pat)
    c, s = np.cos(np.radians(a)), np.sin(np.radians(a))
    near = 0
    for i in range(256):
        v = []
        for e in (0, 1):
            px, py = pat[2 * i + e]
            x, y = px * c - py * s, px * s + py * c
            if min(abs(x - np.floor(x) - 0.5), abs(y - np.floor(y) - 0.5)) < 1e-3:
                near += 1
            v.append(int(img[40 + int(np.rint(y)), 40 + int(np.rint(x))]))
        if near == 0:
            assert ((d[i // 8] >> (i % 8)) & 1) == int(v[0] < v[1]), i
        near = 0


def test_detect_and_compute_invariants(oracle):
    w, h = 640, 360
    img = _img(w, h, 123)
    pat = oracle.orb_random_pattern()
    kps, desc = oracle.orb_detect(img, pat)
    border, lv, sc = oracle.orb_levels(w, h)
    assert len(kps) > 3000 and desc.shape == (len(kps), 32)
    assert np.all(np.diff(kps["octave"]) >= 0)                                                                  # level by level
    for l in range(8):
        m = kps["octave"] == l
        n = int(m.sum())
        assert n <= lv[l][2] or n <= lv[l][2] + 50                                                              # ties of the boundary response may add a few
        if n == 0:
            continue
        x, y = kps["x"][m] / sc[l], kps["y"][m] / sc[l]
        assert x.min() >= 31 - 1e-3 and x.max() < lv[l][0] - 31 + 1e-3 and y.min() >= 31 - 1e-3 and y.max() < lv[l][1] - 31 + 1e-3
        assert np.all(kps["size"][m] == np.float32(31) * sc[l])
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360)) and np.all(kps["class_id"] == -1)
    k2, d2 = oracle.orb_detect(img, None)
    assert d2 is None and np.array_equal(k2, kps)
    # level 0's keypoints are FAST corners of the image itself that survive both rankings
    fast = oracle.fast_detect(img, 10)
    fset = {(int(a), int(b)) for a, b in zip(fast["x"], fast["y"])}
    m0 = kps["octave"] == 0
    assert all((int(a), int(b)) in fset for a, b in zip(kps["x"][m0], kps["y"][m0]))
    # an image shifted by (8, 5) pixels: level-0 keypoints away from the border reappear shifted with identical descriptors
    sh = np.zeros_like(img); sh[5:, 8:] = img[:-5, :-8]
    k3, d3 = oracle.orb_detect(sh, pat)
    idx = {(float(a), float(b)): i for i, (a, b, o) in enumerate(zip(k3["x"], k3["y"], k3["octave"])) if o == 0}
    same = tot = 0
    for i in np.flatnonzero(m0):
        xx, yy = float(kps["x"][i]), float(kps["y"][i])
        if 60 < xx < w - 60 and 60 < yy < h - 60 and (xx + 8, yy + 5) in idx:
            tot += 1; same += int(np.array_equal(desc[i], d3[idx[(xx + 8, yy + 5)]]))
    assert tot > 200 and same == tot
