"""The tail of the stereo loop's stage A as ONE launch (pose.hip: k_stereo_tail): a pair gathers its "after stereo match" set
(VO:569-579), triangulates the rows of that set for its successor (VO:631's points are functions of a row, not of the triangular
match that selects it) and runs extract_3Dpoints (VO:632, VOU:188-232) on the rows of the previous pair's set that its triangular
matches select -- on one small workgroup that keeps a bit per match between its sweeps (extract3d_rows).  What has to hold:
the loop's results are those of the oracle and do not depend on which route the +-3 sigma filter's sums take (UVO_EXTRACT3D_SEQ
forces the ordered chains, MU:35-56), nor on the context's capacity (above 8192 keypoints extract_3Dpoints runs as its any-size
kernel in a launch of its own, reading the same rows through the matches)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WHAT = ("matches_stereo", "matches_tri", "points4d", "good_pts", "good_idx", "inliers")


def fields(r):
    return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers,
            tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr))


def run(uvo, rig, seq, cap, pipelined):
    c = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500), 0, 640, 360, cap)
    try:
        c.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        out = []
        if not pipelined:
            for L, R in seq:
                r = c.stereo_step(L, R, 0.05)
                out.append((fields(r), {w: c.stereo_get(w).copy() for w in WHAT}))
            return out
        c.stereo_set_depth(3)
        sub = 0
        while len(out) < len(seq):
            while sub < len(seq) and sub - len(out) < 3:
                c.stereo_submit(*seq[sub]); sub += 1
            r = c.stereo_collect(0.05)
            out.append((fields(r), {w: c.stereo_get(w).copy() for w in WHAT}))
        return out
    finally:
        c.close()


def same(a, b, skip_first=False):
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], (i, x[0], y[0])
        if skip_first and i == 0:
            continue
        for w in WHAT:
            assert x[1][w].shape == y[1][w].shape and np.array_equal(x[1][w].view(np.uint8), y[1][w].view(np.uint8)), (i, w)


def test_tail_routes_capacities_and_the_oracle():
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    from oracle import pyoracle as po
    scene = synth.Scene(321, 640)
    frames = [synth.stereo_pair(scene, k, 640, 360) for k in range(4)]
    blank = (np.full_like(frames[0][0], 90), np.full_like(frames[0][1], 90))
    # a pair with a few dozen keypoints (a textured patch on a blank image): the launches after it are sized for that count and have to
    # walk the next, rich pair's rows (the tail's roles and the descriptor launch loop over grids smaller than their work)
    sparse = tuple(b.copy() for b in blank)
    for k in (0, 1):
        sparse[k][100:180, 200:300] = frames[1][k][100:180, 200:300]
    seq = [frames[0], frames[1], frames[2], blank, frames[3], sparse, frames[2], frames[1]]       # a gate failure mid-sequence: empty set, VO:727-733
    rig = synth.stereo_rig(640)
    base = run(uvo, rig, seq, 4096, False)
    assert sum(f[0][0] for f in base) >= 4
    try:
        os.environ["UVO_EXTRACT3D_SEQ"] = "1"
        same(run(uvo, rig, seq, 4096, False), base)
        same(run(uvo, rig, seq, 12000, False), base)
    finally:
        os.environ.pop("UVO_EXTRACT3D_SEQ", None)
    same(run(uvo, rig, seq, 12000, False), base)                  # extract_3Dpoints in its own launch
    same(run(uvo, rig, seq, 4096, True), base, skip_first=True)   # (the init pair's lane is reused before its collect)
    same(run(uvo, rig, seq, 12000, True), base, skip_first=True)
    # the oracle (visual_odometry.h:531-739 restated on the CPU)
    p = po.stereo_params(1500)
    ovo = po.StereoVO(p, rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    for i, (L, R) in enumerate(seq):
        o = ovo.step(L, R, 0.05)
        f = base[i][0]
        assert (o.valid, o.initialized, o.n_left, o.n_right, o.n_stereo_matches, o.n_tri_matches, o.n_good3d, o.n_inliers) == f[:8], i
        if o.n_tri_matches:
            assert np.array_equal(ovo.get("points4d").view(np.uint32), base[i][1]["points4d"].view(np.uint32)), i      # both 4 x T
            assert np.array_equal(ovo.get("good_idx"), base[i][1]["good_idx"]), i
            assert np.array_equal(ovo.get("good_pts").view(np.uint64), base[i][1]["good_pts"].view(np.uint64)), i
            assert np.array_equal(ovo.get("inliers"), base[i][1]["inliers"]), i
