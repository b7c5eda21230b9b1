"""Which host threads of a rank are busy while the pipeline runs: python tools/probe/thread_cpu.py [steps] (env as bench.py: UVO_STAGE_B,
UVO_WORKER_WAIT, UVO_CPU_BUDGET).  Per thread: name (comm), CPU seconds per wall second over the timed loop."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(1)
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
ctx.stereo_set_depth(6)
order = [0, 1, 2, 3, 2, 1]
CLK = os.sysconf("SC_CLK_TCK")


def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (name, (int(rest[11]) + int(rest[12])) / CLK)
        except OSError:
            pass
    return out


def run(n):
    sub = 0
    for i in range(n):
        while sub < n and sub - i < 6:
            ctx.stereo_submit(*dev[order[sub % 6]]); sub += 1
        ctx.stereo_collect(0.05)


run(60)
a = snap(); t0 = time.perf_counter()
run(steps)
dt = time.perf_counter() - t0; b = snap()
print("policy:", ctx.host_policy() if hasattr(ctx, "host_policy") else "", "| %.1f pairs/s" % (steps / dt))
rows = sorted(((b[t][1] - a.get(t, (None, 0.0))[1]) / dt, b[t][0], t) for t in b)
tot = 0.0
for busy, name, t in reversed(rows):
    tot += busy
    if busy >= 0.005:
        print("  %-20s tid %-8d %.3f" % (name, t, busy))
print("  total %.3f busy threads" % tot)
ctx.close()
