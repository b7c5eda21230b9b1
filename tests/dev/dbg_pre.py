import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import ergo_uvo_amd as uvo
from oracle import pyoracle as po
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_preproc import _rgb, _cam
ctx = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
img = _rgb(360, 640, 21)
K, d, newK = _cam(320, 180)
want = po.get_image(img, 320, K, d, newK, True, 3)
a = ctx.get_image(img, 320, K, d, newK, True, 3)
print("host/host", (a != want).sum())
t = torch.from_numpy(img).cuda(); torch.cuda.synchronize()
b = ctx.get_image(t, 320, K, d, newK, True, 3)
print("dev/host", (b != want).sum())
c = ctx.get_image(img, 320, K, d, newK, True, 3, device_out=True); torch.cuda.synchronize()
print("host/dev", (c.cpu().numpy() != want).sum())
for cl in (False,):
    w2 = po.get_image(img, 320, K, d, newK, cl, 3)
    a2 = ctx.get_image(img, 320, K, d, newK, cl, 3)
    print("no clahe host/host", (a2 != w2).sum())
ctx.close()
