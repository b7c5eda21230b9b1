"""Known-answer tests of the AKAZE restatement (oracle/o_akaze.c; detect_features' AKAZE branch, VO_utility.cpp:93-98) against the
closed forms of its parts -- PARITY vs OpenCV is UNPINNED; these pin the restatement to the published method (Alcantarilla, Nuevo,
Bartoli, BMVC 2013; FED: Grewenig, Weickert, Bruhn 2010):

  * a FED cycle's step sizes sum to the cycle's stopping time, are d / cos^2(pi (2k + 1) / (4n + 2)) in a permuted order, and the
    evolution times between consecutive levels are reproduced by the cycles;
  * the Scharr pair on a ramp a x + b y gives 32 a and 32 b (kernel sums 16 x a two-pixel difference);
  * the Perona-Malik g2 conductance is 1 on flat regions, 1/2 where |grad L| = k, and decreasing;
  * a diffusion step with unit conductance is step * 2 * (five-point Laplacian), leaves constants alone and conserves the mean away
    from the border;
  * the evolution: 4 octaves x 4 sublevels at 640 x 360 with sigma_size 2, 3, 3, 4 and borders 29, 43, 43, 58; a 100 x 60 image keeps
    one octave;
  * a Gaussian blob is found at its centre (at alternate levels: the cross-scale sweeps compare neighbouring levels only), each keypoint's
    size the diameter 2 x 1.5 x esigma of its level, with a descriptor of 486 bits;
  * rotating the image by 90 degrees maps the keypoints onto the rotated positions with angles turned by 90 degrees and descriptors
    that match across (M-LDB samples a rotated grid)."""
import numpy as np
import pytest


def test_fed_cycle_times(oracle):
    for T in (0.530, 0.2, 1.7, 6.4, 25.0, 103.3):
        tau = oracle.akaze_fed_tau(T)
        n = len(tau)
        assert n == int(np.ceil(np.sqrt(3.0 * T / 0.25 + 0.25) - 0.5 - 1e-8))
        assert abs(float(tau.astype(np.float64).sum()) - T) <= 2e-6 * T * n and np.all(tau > 0)
        scale = 3.0 * T / (0.25 * n * (n + 1))
        want = np.sort(scale * 0.25 / 2.0 / np.cos(np.pi * (2.0 * np.arange(n) + 1.0) / (4.0 * n + 2.0)) ** 2)
        assert np.allclose(np.sort(tau.astype(np.float64)), want, rtol=2e-5)
        if n >= 5:
            assert not np.array_equal(tau, np.sort(tau))                     # the kappa-cycle permutation is not the sorted order (n = 3: kappa 1, the identity)
    lv, es = oracle.akaze_levels(640, 360)
    for i in range(1, len(lv)):                                               # the cycles carry level i - 1 to level i's evolution time
        assert lv[i][5] == len(oracle.akaze_fed_tau(0.5 * (float(es[i]) ** 2 - float(es[i - 1]) ** 2)))


def test_evolution_levels(oracle):
    lv, es = oracle.akaze_levels(640, 360)
    assert len(lv) == 16
    assert [tuple(r[:2]) for r in lv[::4]] == [(640, 360), (320, 180), (160, 90), (80, 45)]
    assert lv[:, 2].tolist() == [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4
    assert lv[:, 3].tolist() == [2, 3, 3, 4] * 4 and lv[:, 4].tolist() == [29, 43, 43, 58] * 4
    assert np.allclose(es, 1.6 * 2.0 ** (np.arange(16) / 4.0), rtol=1e-6)
    lv2, _ = oracle.akaze_levels(100, 60)
    assert len(lv2) == 4                                                       # 50 x 30 is below the smallest octave: one octave only
    lv3, _ = oracle.akaze_levels(641, 363)
    assert [tuple(r[:2]) for r in lv3[::4]] == [(641, 363), (320, 181), (160, 90), (80, 45)]


def test_scharr_on_a_ramp(oracle):
    y, x = np.mgrid[0:40, 0:50].astype(np.float32)
    img = (0.25 * x - 0.125 * y + 3.0).astype(np.float32)
    lx, ly = oracle.akaze_scharr(img, True), oracle.akaze_scharr(img, False)
    assert np.allclose(lx[2:-2, 2:-2], 32 * 0.25, atol=1e-4) and np.allclose(ly[2:-2, 2:-2], 32 * -0.125, atol=1e-4)
    c = np.full((20, 30), 0.7, np.float32)
    assert np.all(oracle.akaze_scharr(c, True) == 0) and np.all(oracle.akaze_scharr(c, False) == 0)      # also at the reflected border


def test_perona_malik_g2(oracle):
    k = 0.05
    g = oracle.akaze_pm_g2(np.array([0.0, k, 0.0, 3 * k], np.float32), np.array([0.0, 0.0, k, 4 * k], np.float32), k)
    assert g[0] == 1.0 and abs(g[1] - 0.5) < 1e-6 and abs(g[2] - 0.5) < 1e-6 and abs(g[3] - 1 / 26.0) < 1e-6
    mags = np.linspace(0, 1, 50, dtype=np.float32)
    gg = oracle.akaze_pm_g2(mags, np.zeros_like(mags), k)
    assert np.all(np.diff(gg) < 0)


def test_diffusion_step(oracle):
    h, w = 24, 31
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    one = np.ones((h, w), np.float32)
    assert np.all(oracle.akaze_nld_step(np.full((h, w), 0.3, np.float32), one, 0.1) == 0)
    L = (0.01 * x * x + 0.02 * y * y).astype(np.float32)                       # Laplacian 0.02 + 0.04
    st = oracle.akaze_nld_step(L, one, 0.05)
    assert np.allclose(st[1:-1, 1:-1], 0.05 * 2 * 0.06, rtol=2e-3)
    rng = np.random.default_rng(0)
    L = np.zeros((h, w), np.float32); L[6:18, 8:22] = rng.random((12, 14), dtype=np.float32)       # support away from the border
    c = (0.2 + rng.random((h, w))).astype(np.float32)
    st = oracle.akaze_nld_step(L, c, 0.2)
    assert abs(float(st.astype(np.float64).sum())) < 1e-5                     # the fluxes cancel pairwise: diffusion conserves the mean
    assert st[0, 0] == 0 and st[0, -1] == 0 and st[-1, 0] == 0 and st[-1, -1] == 0


def _blob_image(w, h, blobs):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 60.0)
    for cx, cy, s, a in blobs:
        img += a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_blob_is_found_at_its_centre(oracle):
    img = _blob_image(320, 240, [(160.0, 120.0, 6.0, 150.0)])
    kps, desc = oracle.akaze_detect(img)
    assert len(kps) >= 1 and desc.shape[1] == 61
    d = np.hypot(kps["x"] - 160.0, kps["y"] - 120.0)
    assert d.max() < 1.0                                                       # one blob: every detection (alternate levels survive the cross-scale sweeps) sits on it
    assert np.all(np.diff(kps["class_id"]) > 0) and np.all(np.diff(kps["size"]) > 0) and np.all(kps["response"] > 0.001)
    assert np.all(kps["size"] == 2 * 1.5 * 1.6 * 2.0 ** (kps["class_id"] / 4.0).astype(np.float32)) or np.allclose(kps["size"], 4.8 * 2.0 ** (kps["class_id"] / 4.0), rtol=1e-6)
    assert np.all(desc[:, 60] < 64)                                            # 486 bits: the last byte holds six
    assert 0 <= kps["angle"].min() and kps["angle"].max() < 360


def test_rotation_by_a_quarter_turn(oracle):
    rng = np.random.default_rng(5)
    blobs = [(rng.uniform(60, 260), rng.uniform(60, 260), rng.uniform(2.5, 7), rng.choice([-1, 1]) * rng.uniform(40, 110)) for _ in range(60)]
    img = _blob_image(320, 320, blobs)
    img = np.clip(img.astype(np.int32) + rng.integers(-2, 3, img.shape), 0, 255).astype(np.uint8)
    k0, d0 = oracle.akaze_detect(img)
    k1, d1 = oracle.akaze_detect(np.ascontiguousarray(np.rot90(img)))          # counter-clockwise: (x, y) -> (y, w - 1 - x)
    assert len(k0) > 40 and abs(len(k1) - len(k0)) <= 0.2 * len(k0)
    das, hams = [], []
    for i in range(len(k0)):
        px, py = k0["y"][i], 319 - k0["x"][i]
        d = np.hypot(k1["x"] - px, k1["y"] - py)
        j = int(np.argmin(d))
        if d[j] < 1.5 and k1["class_id"][j] == k0["class_id"][i]:
            das.append((k0["angle"][i] - k1["angle"][j]) % 360)              # image y points down: a counter-clockwise turn lowers the angle by 90
            hams.append(int(np.unpackbits(d0[i] ^ d1[j]).sum()))
    das, hams = np.array(das), np.array(hams)
    assert len(das) >= 0.9 * len(k0)                                          # the keypoints are the rotated keypoints
    turned = np.abs(das - 90) < 15                                            # (near-isotropic blobs have no stable dominant orientation: a minority strays)
    assert turned.mean() >= 0.8 and abs(float(np.median(das)) - 90) < 1.5
    assert hams[turned].mean() < 0.1 * 486                                    # the rotated grid samples the same cells (observed: 4 % of the bits differ)
