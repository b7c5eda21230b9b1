"""uvo_akaze_detect / uvo_orb_detect timing (image resident in HBM): python tools/prof_binary.py [akaze|orb|both] [steps] [--cpu]
ORB's sampling table here is the one OpenCV's makeRandomPattern draws (the learned bit_pattern_31_ is the integrator's to supply; the
arithmetic is the same)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 20
ctx = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 32768)


def random_pattern(patch=31, npoints=512):                      # orb.cpp makeRandomPattern: cv::RNG(0x34985739).uniform(-patch / 2, patch / 2 + 1)
    state, out = 0x34985739, []
    for _ in range(2 * npoints):
        state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & 0xffffffffffffffff
        out.append(-(patch // 2) + (state & 0xffffffff) % (2 * (patch // 2) + 1))
    return np.array(out, np.int32)


ctx.orb_set_pattern(random_pattern())
for (w, h) in ((1920, 1080), (640, 360)):
    img = synth.stereo_pair(synth.Scene(20250910, w), 0, w, h)[0]
    dimg = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    for name in (("akaze", "orb") if which == "both" else (which,)):
        f = ctx.akaze_detect if name == "akaze" else (lambda im: ctx.orb_detect(im, cap=1 << 15))
        for _ in range(3):
            k, d = f(dimg)
        t0 = time.perf_counter()
        for _ in range(steps):
            k, d = f(dimg)
        dt = (time.perf_counter() - t0) / steps
        line = f"{name}_detect {w}x{h}: {dt*1e3:.3f} ms/frame ({1/dt:.1f} frames/s), {len(k)} keypoints"
        if "--cpu" in sys.argv:
            from oracle import pyoracle as po
            t0 = time.perf_counter()
            ko, do = po.akaze_detect(img, cap=1 << 17) if name == "akaze" else po.orb_detect(img, po.orb_random_pattern())
            tc = time.perf_counter() - t0
            line += f"; CPU oracle (1 thread) {tc*1e3:.0f} ms/frame, identical: {np.array_equal(k.view(np.uint8), ko.view(np.uint8)) and np.array_equal(d, do)}"
        print(line, flush=True)
ctx.close()
