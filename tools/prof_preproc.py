"""get_image timing on the device (frame resident in HBM, result left in HBM): python tools/prof_preproc.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import ergo_uvo_amd as uvo
ctx = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
rng = np.random.default_rng(0)
for (h, w, dw) in ((1080, 1920, 1920), (2160, 3840, 1920), (1080, 1920, 640)):
    img = torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda()
    torch.cuda.synchronize()
    dh = int(h / (w / dw))
    K = np.array([[0.9 * dw, 0, 0.51 * dw], [0, 0.92 * dw, 0.49 * dh], [0, 0, 1.0]]); newK = K.copy(); newK[0, 0] *= 0.93; newK[1, 1] *= 0.93
    d = np.array([-0.21, 0.06, 0.0012, -0.0017])
    for _ in range(5):
        ctx.get_image(img, dw, K, d, newK, True, 8, device_out=True)
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        ctx.get_image(img, dw, K, d, newK, True, 8, device_out=True)
    dt = (time.perf_counter() - t0) / n
    print(f"get_image {w}x{h} -> {dw}x{dh}: {dt*1e3:.3f} ms/frame ({1/dt:.0f} frames/s), input {w*h*3/1e6:.1f} MB")
ctx.close()
