"""Host-side cost of the pipelined stereo path: time spent inside uvo_stereo_submit / uvo_stereo_collect per pair.
python tools/prof_pipeline.py [depth] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene = synth.Scene(synth.SEEDS["C3"], W)
dev = [tuple(torch.from_numpy(x).cuda() for x in synth.stereo_pair(scene, k, W, H)) for k in range(4)]
rig = synth.stereo_rig(W)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387, SURF_OCTAVES_NUMBER=int(os.environ.get("UVO_OCTAVES", "4"))), 0, W, H, 8192)
ctx.stereo_set_depth(depth)
ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
order = [0, 1, 2, 3, 2, 1]
for i in range(8):
    ctx.stereo_step(*dev[order[i % 6]], 0.05)
torch.cuda.synchronize()
ts = tc = 0.0
sub = 0
t0 = time.perf_counter()
for i in range(steps):
    while sub < steps and sub - i < depth:
        a = time.perf_counter(); ctx.stereo_submit(*dev[order[sub % 6]]); ts += time.perf_counter() - a; sub += 1
    a = time.perf_counter(); r = ctx.stereo_collect(0.05); tc += time.perf_counter() - a
dt = time.perf_counter() - t0
print(f"depth {depth}: {steps/dt:.1f} pairs/s, {dt/steps*1e3:.3f} ms/pair; host in submit {ts/steps*1e6:.0f} us/pair, in collect {tc/steps*1e6:.0f} us/pair; valid {r.valid}")
ctx.close()
