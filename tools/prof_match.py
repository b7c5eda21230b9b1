"""Matcher micro-run for rocprofv3: python tools/prof_match.py [n] (random unit 64-D descriptors, n x n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ergo_uvo_amd as uvo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(5)
a = rng.normal(size=(n, 64)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
b = (a[rng.permutation(n)] + 0.15 * rng.normal(size=(n, 64)).astype(np.float32)); b /= np.linalg.norm(b, axis=1, keepdims=True)
ctx = uvo.Context(uvo.Params.stereo(), 0, 640, 480, 8192)
for _ in range(6):
    idx, dist = ctx.knn_match(a, b.astype(np.float32))
print(idx[:3], dist[:3])
ctx.close()
