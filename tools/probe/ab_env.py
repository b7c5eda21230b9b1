"""A/B of library settings that are read from the environment when a context is created (UVO_WORKER_WAIT, UVO_PNP_PRIORITY,
UVO_MAX_B, UVO_A_OVERLAP, ...) inside ONE process, on the C3 workload: for every variant a fresh context, then
  * the driver's form: fence, 20 pairs through submit/collect at depth 6, fence -- repeated, median and best pairs/s
  * the long form: 600 pairs
  * the synchronous step's latency (median of 100)
  * busy host threads (process CPU seconds per wall second) during the long form
python tools/probe/ab_env.py "NAME=VAL,NAME2=VAL" "NAME=VAL2" ...      ("-" = library defaults)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth

W, H, DEPTH = 1920, 1080, int(os.environ.get("AB_DEPTH", "6"))
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
order = [0, 1, 2, 3, 2, 1]


def piped(ctx, n, start):
    sub = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        while sub < n and sub - i < DEPTH:
            ctx.stereo_submit(*dev[order[(start + sub) % 6]]); sub += 1
        ctx.stereo_collect(0.05)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


variants = sys.argv[1:] or ["-"]
touched = set()
for v in variants:
    for k in touched:
        os.environ.pop(k, None)
    if v != "-":
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val; touched.add(k)
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
    ctx.stereo_set_depth(DEPTH)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    for i in range(2):
        ctx.stereo_step(*dev[order[i]], 0.05)
    piped(ctx, 60, 2)
    short = sorted(piped(ctx, 20, 2 + 60 + 20 * r) for r in range(15))
    c0 = time.process_time(); t0 = time.perf_counter()
    long_ = piped(ctx, 600, 2)
    busy = (time.process_time() - c0) / (time.perf_counter() - t0)
    lat = []
    for i in range(100):
        a = time.perf_counter(); ctx.stereo_step(*dev[order[i % 6]], 0.05); lat.append((time.perf_counter() - a) * 1e3)
    lat.sort()
    print("%-44s 20-step median %.0f best %.0f | 600-step %.0f pairs/s | sync median %.3f ms p95 %.3f | busy host threads %.2f"
          % (v, short[len(short) // 2], short[-1], long_, lat[50], lat[95], busy), flush=True)
    ctx.close()
