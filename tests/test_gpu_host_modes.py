"""The host side of the stereo pipeline has several ways to wait and two ways to drive the PnP stage of a pipelined pair
(ergo_uvo_amd/csrc/uvo_ctx.h: worker_wait, stage_b_mode; include/uvo_hip.h: uvo_ctx_host_policy) -- chosen from the process's CPU
budget so that a rank of an 8-GPU run lives on its 2-CPU share of the box's quota.  None of them may change a result: every mode
must return, pair for pair and bit for bit, what the synchronous uvo_stereo_step returns, through the gate failures of the loop
(visual_odometry.h:556/567/626/634/665) and through the cases in which the device-driven RANSAC round cannot decide a pair and the
collecting thread runs the host-driven stage (visual_odometry.h:647-676)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = [
    ("spin", {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "spin"}, {"wait": "poll", "stage_b": "worker"}),
    ("sleep", {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "sleep"}, {"wait": "timed-sleep+poll", "stage_b": "worker"}),
    ("block-all", {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "block-all"}, {"wait": "interrupt", "stage_b": "worker"}),
    ("device", {"UVO_STAGE_B": "device"}, {"stage_b": "device"}),
    ("device, interrupt waits", {"UVO_STAGE_B": "device", "UVO_WORKER_WAIT": "block-all"}, {"wait": "interrupt", "stage_b": "device"}),
    ("two CPUs, the library's own choice", {"UVO_CPU_BUDGET": "2"}, {"wait": "timed-sleep+poll", "stage_b": "device", "cpu_budget": 2.0}),
    ("forty CPUs, the library's own choice", {"UVO_CPU_BUDGET": "40"}, {"stage_b": "worker"}),
]
ENV_KEYS = ("UVO_STAGE_B", "UVO_WORKER_WAIT", "UVO_CPU_BUDGET")


def _fields(r):
    return (r.valid, r.initialized, r.n_left, r.n_right, r.n_stereo_matches, r.n_tri_matches, r.n_good3d, r.n_inliers,
            tuple(r.rvec), tuple(r.tvec), tuple(r.t_prev_curr), tuple(r.velocity))


def _piped(ctx, rig, seq, depth):
    ctx.stereo_set_depth(depth)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    out, inl, sub = [], [], 0
    for i in range(len(seq)):
        while sub < len(seq) and sub - i < max(depth, 1):
            ctx.stereo_submit(*seq[sub]); sub += 1
        out.append(_fields(ctx.stereo_collect(0.05)))
        inl.append(ctx.stereo_get("inliers").copy())
    return out, inl


@pytest.fixture(scope="module")
def cases(scene_small):
    blank = (np.full_like(scene_small[0][0], 90), np.full_like(scene_small[0][1], 90))
    half = (scene_small[1][0], blank[1])
    gates = [scene_small[0], scene_small[1], blank, scene_small[2], scene_small[1], half, scene_small[0], scene_small[1], scene_small[2], scene_small[1]]
    plain = [scene_small[k] for k in (0, 1, 2, 1, 0, 1, 2, 1, 0)]
    return [("gate failures in the middle of a sequence", {}, gates),
            ("0.03-px threshold: the scan reaches past the round's 64 hypotheses", dict(REPROJECTION_ERROR_THRESHOLD=0.03), plain),
            ("30 iterations allowed", dict(ITERATIONS_COUNT=30), plain),
            ("a 3-D point gate that never opens", dict(MIN_NUM_3DPOINTS=100000), plain)]


@pytest.fixture(scope="module")
def want(cases, oracle):
    """The synchronous step's results of every case (default host policy), checked against the oracle."""
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    saved = {k: os.environ.pop(k, None) for k in ENV_KEYS}
    out = []
    try:
        for name, kw, seq in cases:
            ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, **kw), 0, 640, 360, 4096)
            op = oracle.stereo_params(1500)
            for k, v in kw.items():
                setattr(op, k, v)
            ovo = oracle.StereoVO(op, rig.K_left, rig.K_right, rig.R_right, rig.t_right)
            try:
                ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
                res, inl = [], []
                for L, R in seq:
                    r = ctx.stereo_step(L, R, 0.05)
                    o = ovo.step(L, R, 0.05)
                    assert _fields(r)[:8] == _fields(o)[:8], (name, _fields(r)[:8], _fields(o)[:8])
                    assert np.array_equal(ctx.stereo_get("inliers"), ovo.get("inliers")), name
                    res.append(_fields(r)); inl.append(ctx.stereo_get("inliers").copy())
                out.append((res, inl))
            finally:
                ovo.close(); ctx.close()
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    return out


@pytest.mark.parametrize("label,env,policy", MODES, ids=[m[0] for m in MODES])
def test_every_host_mode_returns_the_synchronous_steps_results(cases, want, label, env, policy):
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    rig = synth.stereo_rig(640)
    saved = {k: os.environ.pop(k, None) for k in ENV_KEYS}
    os.environ.update(env)
    try:
        for (name, kw, seq), (res, inl) in zip(cases, want):
            ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=1500, **kw), 0, 640, 360, 4096)
            try:
                for depth in (4, 2, 6):
                    got, ginl = _piped(ctx, rig, seq, depth)
                    pol = ctx.host_policy()
                    for k, v in policy.items():
                        assert pol[k] == v, (label, pol)
                    assert pol["depth"] == depth
                    assert got == res, (label, name, depth)
                    for a, b in zip(ginl, inl):
                        assert np.array_equal(a, b), (label, name, depth)
            finally:
                ctx.close()
    finally:
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def test_a_device_driven_rank_keeps_about_one_host_thread_busy():
    """What the mode is for: at C3 with six pairs in flight the process burns about two CPU seconds per second with the PnP round on the
    device -- the polling submit / collect thread and one thread of the HIP runtime that the pipeline's completion signals keep awake
    (tools/probe/thread_cpu.py: 0.99 + 0.96; the lane workers sleep) -- against 7-9 when every lane's worker polls.  The bound leaves
    room for the runtime's other helpers (1.95-2.04 observed)."""
    import time
    import torch
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth
    import bench
    W, H = bench.WIDTH, bench.HEIGHT
    scene = synth.Scene(synth.SEEDS["C3"], W)
    dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(3)]
    rig = synth.stereo_rig(W)
    saved = {k: os.environ.pop(k, None) for k in ENV_KEYS}
    busy = {}
    try:
        for mode in ("device", "worker"):
            os.environ["UVO_STAGE_B"] = mode
            os.environ["UVO_WORKER_WAIT"] = "spin"
            ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=bench.MIN_HESSIAN_C3), 0, W, H, 8192)
            try:
                ctx.stereo_set_depth(6)
                ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
                order, n, sub = [0, 1, 2, 1], 400, 0
                for i in range(n):
                    if i == 100:
                        c0, t0 = time.process_time(), time.perf_counter()
                    while sub < n and sub - i < 6:
                        ctx.stereo_submit(*dev[order[sub % 4]]); sub += 1
                    r = ctx.stereo_collect(0.05)
                    assert r.valid == (1 if i else 0)
                busy[mode] = (time.process_time() - c0) / (time.perf_counter() - t0)
            finally:
                ctx.close()
    finally:
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    assert busy["device"] <= 2.4, busy
    assert busy["worker"] >= busy["device"] + 1.0, busy
