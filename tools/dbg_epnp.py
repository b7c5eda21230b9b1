import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = '''
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth
rig = synth.stereo_rig(1920)
rng = np.random.default_rng(5)
n=2000
X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.5, 6, n)], 1)
Rt, tt = synth.true_relative_motion()
Y = X @ Rt.T + tt
K = rig.K_left
x = ((Y[:, :2] / Y[:, 2:]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
c = uvo.Context(uvo.Params.stereo(), 0, 1920, 1080, 8192)
c.solvePnPRansac(X, x, K)
c.timing_enable(True); c.timing_reset()
for _ in range(5): c.solvePnPRansac(X, x, K)
t = c.timing()
print(os.environ.get("UVO_DBG_PHASE"), {k: round(v[0]/max(v[1],1),4) for k,v in t.items() if v[1]})
'''
for ph in [0,1,2,3,4,5,6,99]:
    env = dict(os.environ, UVO_DBG_PHASE=str(ph))
    subprocess.run([sys.executable, "-c", code], env=env)
