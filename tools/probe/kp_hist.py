"""Histogram of the descriptor window sizes of the bench scene's keypoints: python tools/probe/kp_hist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
W, H = 1920, 1080
scene = synth.Scene(synth.SEEDS["C3"], W)
L, R = synth.stereo_pair(scene, 0, W, H)
ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
kps, desc = ctx.detect_features(L)
size = np.array([k[2] if not hasattr(k, "size") else k.size for k in kps]) if not isinstance(kps, np.ndarray) else kps["size"]
win = (21 * (size.astype(np.float32) * np.float32(1.2) / np.float32(9.0))).astype(int)
print("n", len(win), "big (>128)", int((win > 128).sum()), "taps small", int((win[win <= 128].astype(np.int64) ** 2).sum()), "taps big", int((win[win > 128].astype(np.int64) ** 2).sum()))
for lo, hi in [(0, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]:
    m = (win > lo) & (win <= hi)
    print("win (%d,%d]: %d kps, mean %.0f" % (lo, hi, int(m.sum()), win[m].mean() if m.any() else 0))
ctx.close()
