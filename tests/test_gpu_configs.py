"""Parity of the HIP path against the CPU oracle AT THE BASELINE.json CONFIGURATIONS (sizes, seeds and thresholds of
SURVEY.md 8(d)), synchronous and with six pairs / frames in flight:

  C1 substitute  mono 640x480, the shipped LMedS parameters (mono_VO_parameters.yaml: 50 / 0.7 / 0.1), seed 20250904
                 (the shipped bag is absent: README.md:78-80) -- E and H branches
  C2             stereo 1280x720, ~1500 keypoints (min_hessian 5685), seed 20250905
  C3 (headline)  stereo 1920x1080, ~3000 keypoints (min_hessian 6387), seed 20250906, cap 8192 as bench.py
  C4             mono 1920x1080 + range, 3000 keypoints at frame 0 (min_hessian 6456), RANSAC for E and H, seed 20250907;
                 the sequence has full-parallax frames (essential branch) and quarter-step frames (median displacement
                 < DISTANCE: homography branch), so both estimators are run and scored at 1080p

Bit-exact keypoints, descriptors, match lists, points4d, good_idx, inlier sets / masks; poses to 1e-4 relative
(north_star).  PARITY vs OpenCV itself is UNPINNED: the oracle is this repository's restatement.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEREO_FIELDS = ("valid", "initialized", "n_left", "n_right", "n_stereo_matches", "n_tri_matches", "n_good3d", "n_inliers")
MONO_FIELDS = ("published", "valid", "initialized", "used_essential", "success", "n_kps", "n_matches", "n_inliers", "n_good3d", "n_front")
POSE_TOL = 1e-4          # north_star: pose within 1e-4 relative (Frobenius)


def _kps_equal(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        av, bv = a[f], b[f]
        if av.dtype.kind == "f":
            assert np.array_equal(av.view(np.uint32), bv.view(np.uint32)), f
        else:
            assert np.array_equal(av, bv), f


def _matches_equal(a, b):
    assert len(a) == len(b)
    assert np.array_equal(a["queryIdx"], b["queryIdx"]) and np.array_equal(a["trainIdx"], b["trainIdx"])
    assert np.array_equal(a["distance"].view(np.uint32), b["distance"].view(np.uint32))


def _rel(a, b):
    a, b = np.array(list(a), np.float64), np.array(list(b), np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _oracle_stereo_run(oracle, params, rig, seq):
    """The oracle's result and every intermediate of each pair of `seq`."""
    ovo = oracle.StereoVO(params, rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    want = []
    for L, R in seq:
        o = ovo.step(L, R, 0.05)
        want.append((o, {k: ovo.get(k) for k in ("kps_left", "kps_right", "desc_left", "desc_right", "matches_stereo", "matches_tri",
                                                  "points4d", "good_pts", "good_idx", "inliers")}))
    ovo.close()
    return want


def _check_stereo_pair(ctx, r, want, k):
    o, s = want
    for f in STEREO_FIELDS:
        assert getattr(r, f) == getattr(o, f), (k, f, getattr(r, f), getattr(o, f))
    _kps_equal(ctx.stereo_get("kps_left"), s["kps_left"])
    _kps_equal(ctx.stereo_get("kps_right"), s["kps_right"])
    assert np.array_equal(ctx.stereo_get("desc_left").view(np.uint32), s["desc_left"].view(np.uint32))
    assert np.array_equal(ctx.stereo_get("desc_right").view(np.uint32), s["desc_right"].view(np.uint32))
    _matches_equal(ctx.stereo_get("matches_stereo"), s["matches_stereo"])
    if r.initialized:
        _matches_equal(ctx.stereo_get("matches_tri"), s["matches_tri"])
        assert np.array_equal(ctx.stereo_get("points4d").view(np.uint32), s["points4d"].view(np.uint32))
        assert np.array_equal(ctx.stereo_get("good_idx"), s["good_idx"])
        assert np.array_equal(ctx.stereo_get("good_pts").view(np.uint64), s["good_pts"].view(np.uint64))
        assert np.array_equal(ctx.stereo_get("inliers"), s["inliers"])                      # bit-exact inlier set
        for a, b in ((r.rvec, o.rvec), (r.tvec, o.tvec), (r.t_prev_curr, o.t_prev_curr), (r.velocity, o.velocity)):
            assert _rel(a, b) <= POSE_TOL, (k, list(a), list(b))


def _run_stereo_config(oracle, seed, W, H, min_hessian, order, cap=8192, depth=6, min_kpts=0):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    scene = synth.Scene(seed, W)
    frames = {k: synth.stereo_pair(scene, k, W, H) for k in sorted(set(order))}
    seq = [frames[k] for k in order]
    rig = synth.stereo_rig(W)
    want = _oracle_stereo_run(oracle, oracle.stereo_params(min_hessian), rig, seq)
    assert sum(o.valid for o, _ in want) == len(seq) - 1 and want[0][0].n_left >= min_kpts
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=min_hessian), 0, W, H, cap)
    try:
        # synchronous uvo_stereo_step
        ctx.stereo_set_depth(1)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        for k, (L, R) in enumerate(seq):
            _check_stereo_pair(ctx, ctx.stereo_step(L, R, 0.05), want[k], ("sync", k))
        # `depth` pairs in flight: every collected pair's intermediates are read from its lane before the lane is reused
        ctx.stereo_set_depth(depth)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        ctx.stereo_submit(*seq[0])
        _check_stereo_pair(ctx, ctx.stereo_collect(0.05), want[0], ("piped-init", 0))
        sub = 1
        for i in range(1, len(seq)):
            while sub < len(seq) and sub - i < depth:
                ctx.stereo_submit(*seq[sub]); sub += 1
            _check_stereo_pair(ctx, ctx.stereo_collect(0.05), want[i], ("piped", i))
    finally:
        ctx.close()
    return want


def test_c3_headline_stereo_1080p_parity(oracle):
    """BASELINE configs[2]: the configuration bench.py times (same seed, threshold, cap and pipeline depth)."""
    from ergo_uvo_amd import synth
    want = _run_stereo_config(oracle, synth.SEEDS["C3"], 1920, 1080, 6387, [0, 1, 2, 3, 2, 1, 0, 1, 2, 3], cap=8192, depth=6, min_kpts=2900)
    assert abs(want[0][0].n_left - 3000) <= 90                          # 3000 +- 3 % at frame 0


def test_c2_stereo_720p_parity(oracle):
    from ergo_uvo_amd import synth
    want = _run_stereo_config(oracle, synth.SEEDS["C2"], 1280, 720, 5685, [0, 1, 2, 3, 2, 1, 0, 1], cap=8192, depth=6, min_kpts=1400)
    assert abs(want[0][0].n_left - 1500) <= 45


def _run_mono_config(oracle, seed, W, H, params_kw, oparams, ks, depth=6):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    scene = synth.Scene(seed, W)
    frames = {k: synth.mono_frame(scene, k, W, H) for k in sorted(set(ks))}
    seq = [frames[k] for k in ks]
    rig = synth.stereo_rig(W)
    R0, C0 = synth.camera_pose(0)
    rng = scene.depth_at_center(C0, R0)
    ovo = oracle.MonoVO(oparams, rig.K_left)
    want = []
    for img in seq:
        o = ovo.step(img, rng, 0.2)
        want.append((o, ovo.get("kps"), ovo.get("matches"), ovo.get("mask"), ovo.get("good_pts")))
    ovo.close()

    def check(ctx, r, w, tag):
        o, okps, om, omask, ogood = w
        for f in MONO_FIELDS:
            assert getattr(r, f) == getattr(o, f), (tag, f, getattr(r, f), getattr(o, f))
        _kps_equal(ctx.mono_get("kps"), okps)
        if r.published:
            _matches_equal(ctx.mono_get("matches"), om)
            assert np.array_equal(ctx.mono_get("mask"), omask)                              # bit-exact inlier mask
            g = ctx.mono_get("good_pts")
            assert g.shape == ogood.shape and np.array_equal(g.view(np.uint64), ogood.view(np.uint64))
            for a, b in ((r.R, o.R), (r.t, o.t), (r.velocity, o.velocity), ([r.SF], [o.SF])):
                assert _rel(a, b) <= POSE_TOL, (tag, list(a), list(b))

    ctx = uvo.Context(uvo.Params.mono(**params_kw), 0, W, H, 8192)
    try:
        ctx.mono_set_camera(rig.K_left)
        for k, img in enumerate(seq):
            check(ctx, ctx.mono_step(img, rng, 0.2), want[k], ("sync", k))
        ctx.mono_reset()
        ctx.stereo_set_depth(depth)
        sub = 0
        for i in range(len(seq)):
            while sub < len(seq) and sub - i < depth:
                ctx.mono_submit(seq[sub], rng); sub += 1
            check(ctx, ctx.mono_collect(0.2), want[i], ("piped", i))
    finally:
        ctx.close()
    return want


def test_c4_mono_1080p_ransac_e_and_h_parity(oracle):
    """BASELINE configs[3], the 1.0-px VARIANT (thresholds 1.0 / 1.0 / 3.0 instead of the shipped 0.1 / 0.1 / 0.1, so that the
    homography branch produces valid poses on this synthetic scene): RANSAC for both estimators; two-step frames take the
    essential branch, quarter-step frames the homography branch (select_estimation_method, VOU:725-748).  The contract's own
    thresholds are test_c4_mono_1080p_contract_thresholds_parity below."""
    from ergo_uvo_amd import synth
    kw = dict(SURF_MIN_HESSIAN=6456, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8, ESSENTIAL_THRESHOLD=1.0,
              HOMOGRAPHY_THRESHOLD=1.0, REPROJECTION_TOLERANCE=3.0)
    op = oracle.mono_params(6456, method=8)
    op.ESSENTIAL_THRESHOLD = 1.0; op.HOMOGRAPHY_THRESHOLD = 1.0; op.REPROJECTION_TOLERANCE = 3.0
    ks = [-2, 0, 2, 0, 0.25, 0.5, 0.25, 2, 4]
    want = _run_mono_config(oracle, synth.SEEDS["C4"], 1920, 1080, kw, op, ks)
    res = [w[0] for w in want]
    assert abs(res[1].n_kps - 3000) <= 90                               # 3000 +- 3 % at frame 0
    pub = [r for r in res if r.published]
    assert len(pub) == len(ks) - 1 and all(r.valid for r in pub)
    assert sum(r.used_essential for r in pub) >= 4 and sum(1 - r.used_essential for r in pub) >= 3       # both branches scored


def test_c4_mono_1080p_contract_thresholds_parity(oracle):
    """BASELINE configs[3] at the contract's parameters ("the mono column but with methods = 8"):
    essential_threshold 0.1, homography_threshold 0.1, reprojection_tolerance 0.1 (uvo/config/mono_VO_parameters.yaml:21, 26,
    30), RANSAC for both estimators.  At 0.1 px the adaptive iteration count stays in the hundreds to thousands, so the
    five-point and the DLT kernels run whole rounds of hypotheses.  On this synthetic scene the two-step frames (parallax) are
    solved by the essential branch; the quarter-step frames start on the homography branch (median displacement < DISTANCE),
    fail VO_utility.cpp:164's inlier-fraction gate, switch to the essential matrix and fail it too after recoverPose's
    cheirality filter -- "BOTH METHODS FAILED", success = 0, the node keeps its previous motion: there both RANSACs are run and
    scored on the same frame, and what is compared is the masks and the gate ladder."""
    from ergo_uvo_amd import synth
    kw = dict(SURF_MIN_HESSIAN=6456, ESSENTIAL_OUTLIER_METHOD=8, HOMOGRAPHY_OUTLIER_METHOD=8)
    op = oracle.mono_params(6456, method=8)
    for f, v in (("ESSENTIAL_THRESHOLD", 0.1), ("HOMOGRAPHY_THRESHOLD", 0.1), ("REPROJECTION_TOLERANCE", 0.1)):
        assert getattr(op, f) == v                                      # the shipped values
    ks = [-2, 0, 2, 0, 0.25, 0.5, 0.25, 2, 4]
    want = _run_mono_config(oracle, synth.SEEDS["C4"], 1920, 1080, kw, op, ks)
    res = [w[0] for w in want]
    pub = [r for r in res if r.published]
    assert len(pub) == len(ks) - 1
    assert sum(r.valid for r in pub) >= 5 and sum(1 - r.success for r in pub) >= 3      # parallax frames solved; quarter-step frames: both methods scored, both gated out
    assert all(r.n_inliers >= 1000 for r in pub)


def test_c1_substitute_mono_640x480_shipped_lmeds_parity(oracle):
    """BASELINE configs[0] cannot be run (bag and OpenCV absent); its substitute: 640x480, the shipped mono parameters."""
    from ergo_uvo_amd import synth
    ks = [0, 2, 4, 4.25, 4.5, 6, 4, 2]
    want = _run_mono_config(oracle, synth.SEEDS["C1"], 640, 480, {}, oracle.mono_params(), ks, depth=3)
    pub = [w[0] for w in want if w[0].published]
    assert len(pub) == len(ks) - 1 and all(r.valid for r in pub)
    assert any(r.used_essential for r in pub) and any(not r.used_essential for r in pub)


def _bench_ranks_vs_single_gpu_streams(tmp_path, n, extra, base_seed_key):
    """Runs `python bench.py --gpus n ...` as a child (bench.py starts its own ranks; none of them execs) and checks the gathered
    record of rank r against the synchronous single-GPU result of stream r, bit for bit.  Returns the JSON line."""
    import json
    import os
    import subprocess
    import sys
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth, multirank
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    steps, warm, frames = 6, 3, 2
    rec_path = str(tmp_path / "records.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "UVO_RDZV_FILE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warm),
                        "--frames", str(frames), "--no-cpu-baseline", "--dump-records", rec_path] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1100)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == n and line["scaling"] == "weak" and line["config"]["valid_steps"] == steps
    rec = np.load(rec_path)                                   # the records of the median timed block
    assert rec.shape == (n, steps, multirank.RECORD_WIDTH)
    warm = line["pairs_before_median_block"]                  # pairs the stream had seen before that block (init, warm-up, earlier blocks)
    assert warm >= 2 + 2 * 2 and line["blocks"] >= 1 and len(line["block_values"]) == line["blocks"] and len(line["per_rank_value"]) == n
    W, H = bench.WIDTH, bench.HEIGHT
    rig = synth.stereo_rig(W)
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=bench.MIN_HESSIAN_C3), 0, W, H, 8192)
    try:
        for r in range(n):
            seed = synth.SEEDS[base_seed_key] if base_seed_key == "C3" else multirank.stream_seed(synth.SEEDS["C5"], r)
            scene = synth.Scene(seed, W)
            pairs = [synth.stereo_pair(scene, k, W, H) for k in range(frames)]
            ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
            order = bench.ping_pong(frames)
            want = np.zeros((warm + steps, multirank.RECORD_WIDTH))
            for i in range(warm + steps):
                multirank.fill_record(want, i, r, max(i - warm, 0), ctx.stereo_step(*pairs[next(order)], 0.05))
            assert np.array_equal(rec[r].view(np.uint64), want[warm:].view(np.uint64)), r
    finally:
        ctx.close()
    if n > 1:
        assert not np.array_equal(rec[0, :, 7:10], rec[1, :, 7:10])          # different streams
    return line


def test_c5_self_launched_two_ranks_match_single_gpu_streams(tmp_path):
    """BASELINE configs[4] rehearsed on the one device of the GPU box: `python bench.py --gpus 2` starts its two ranks itself
    (no torchrun); both ranks share device 0 and gather over gloo (RCCL refuses two ranks on one device), everything else --
    one independent stream per rank, seed 20250910 + rank, one gather of the pose records at the end of the timed region -- is
    the path the 8-GPU run takes."""
    line = _bench_ranks_vs_single_gpu_streams(tmp_path, 2, ["--backend", "gloo", "--share-devices"], "C5")
    assert line["host"]["collectives"].startswith("gloo") and line["host"]["pinned"]


def test_c5_one_rank_rccl_communicator_runs_the_collectives(tmp_path):
    """The RCCL leg of configs[4] on the one GPU there is: `--gpus 1 --backend nccl --force-dist` builds a communicator of one
    rank on the device and runs the very calls of the N-rank path -- all_gather_into_tensor of the device-resident pose
    records, all_reduce(MAX) of the time, the barriers -- so the 8-GPU run is not the first time they execute.  The gathered
    record must be the local one bit for bit (bench.py asserts it too) and equal to the synchronous single-GPU run."""
    line = _bench_ranks_vs_single_gpu_streams(tmp_path, 1, ["--backend", "nccl", "--force-dist"], "C3")
    assert line["host"]["collectives"].startswith("nccl") and "1 rank" in line["host"]["collectives"] and line["rccl_ranks_seen"] == 1
    assert line["host"]["pin_matches_opened_device"] in (True, None)                   # the rank sits next to the card it opened


def test_c5_five_ranks_on_one_device_match_single_gpu_streams(tmp_path):
    """As many ranks as the GPU box allows next to this process (its guard admits six GPU processes): five self-launched ranks,
    seeds 20250910..14, depth 2, share device 0 over gloo; each gathered stream must equal its single-GPU run.  (The eight-rank
    form of the same path runs on CPU in tests/test_multirank.py.)"""
    line = _bench_ranks_vs_single_gpu_streams(tmp_path, 5, ["--backend", "gloo", "--share-devices", "--depth", "2"], "C5")
    assert line["host"]["cores_of_this_rank"] >= 1
    # `--gpus 8 --backend gloo --share-devices --depth 2` is NOT run here: the GPU box's process guard admits six processes on the
    # card at once (this test process + 5 ranks), and a ninth would have the run killed ("process guard").
    assert line["ranks_seen"] == [0, 1, 2, 3, 4] and line["rccl_ranks_seen"] == 0 and len(set(line["host"]["numa_nodes_of_ranks"])) == 1
    assert sum(line["host"]["cores_of_ranks"]) <= line["host"]["cores_visible"]         # the five ranks split the card's NUMA node


@pytest.mark.parametrize("stream", range(8))
def test_c5_every_stream_against_the_oracle(oracle, stream):
    """BASELINE configs[4] under the ORACLE (the self-launched-ranks tests above compare gathered records with a synchronous HIP run of
    the same stream -- plumbing; this one is the parity check): each of the eight streams (seeds 20250910 + stream), the
    init pair and four more, synchronously and with six pairs in flight, every intermediate of every pair against oracle.StereoVO bit
    for bit (keypoints, descriptors, both match lists, points4d, good_idx / good_pts, inlier sets; poses to 1e-4).  One device, one
    stream after the other, as rank r of the 8-GPU run would see stream r (visual_odometry.h:723-733: nothing is shared between streams)."""
    from ergo_uvo_amd import synth, multirank
    import bench
    want = _run_stereo_config(oracle, multirank.stream_seed(synth.SEEDS["C5"], stream), bench.WIDTH, bench.HEIGHT, bench.MIN_HESSIAN_C3,
                              [0, 1, 2, 1, 0], cap=8192, depth=6, min_kpts=2000)
    assert all(o.n_inliers >= 100 for o, _ in want[1:])


def test_submit_blocks_for_at_most_a_detector_stage():
    """uvo_stereo_submit paces the pipeline on the calling thread: before it queues a pair's kernels it waits (polling) for the end
    of the stage A submitted two pairs earlier, so a "submit" may hold its caller for up to about one
    stage A -- ~0.5 ms at C3 -- and never for a pair's whole latency.  A node that calls it from a 20 Hz loop
    (visual_odometry.h:526-530) has 50 ms per frame.  Pinned here: over a pipelined C3 run no submit call takes longer than 5 ms,
    the median is below 1 ms, and collect() after a full pipeline returns the oldest pair without a long wait."""
    import time
    import torch
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    import bench
    W, H = bench.WIDTH, bench.HEIGHT
    scene = synth.Scene(synth.SEEDS["C3"], W)
    dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(3)]
    rig = synth.stereo_rig(W)
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=bench.MIN_HESSIAN_C3), 0, W, H, 8192)
    try:
        ctx.stereo_set_depth(6)
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
        order = [0, 1, 2, 1]
        t_sub = []
        sub = 0
        n = 120
        for i in range(n):
            while sub < n and sub - i < 6:
                a = time.perf_counter(); ctx.stereo_submit(*dev[order[sub % 4]]); t_sub.append(time.perf_counter() - a); sub += 1
            ctx.stereo_collect(0.05)
        t = np.array(t_sub[12:]) * 1e3          # after the synchronous init pair and the first fill
        assert t.max() < 5.0, t.max()
        assert np.median(t) < 1.0, np.median(t)
    finally:
        ctx.close()
