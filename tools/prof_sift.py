"""uvo_sift_detect timing (image resident in HBM): python tools/prof_sift.py [steps]; CPU oracle timed beside it with --cpu."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
ctx = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
for (w, h) in ((1920, 1080), (640, 360)):
    img = synth.stereo_pair(synth.Scene(20250910, w), 0, w, h)[0]
    dimg = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        k, d = ctx.sift_detect(dimg)
    t0 = time.perf_counter()
    for _ in range(steps):
        k, d = ctx.sift_detect(dimg)
    dt = (time.perf_counter() - t0) / steps
    line = f"sift_detect {w}x{h}: {dt*1e3:.3f} ms/frame ({1/dt:.1f} frames/s), {len(k)} keypoints"
    if "--cpu" in sys.argv:
        from oracle import pyoracle as po
        t0 = time.perf_counter(); ko, do = po.sift_detect(img); tc = time.perf_counter() - t0
        line += f"; CPU oracle (1 thread) {tc*1e3:.0f} ms/frame, identical: {np.array_equal(k.view(np.uint8), ko.view(np.uint8)) and np.array_equal(d, do)}"
    print(line, flush=True)
ctx.close()
