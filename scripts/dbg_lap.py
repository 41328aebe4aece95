import importlib, numpy as np, sys
sys.path.insert(0, '.')
bts = importlib.import_module("bot-sort-onnx-tensorrt_amd")
from oracle import bts_oracle as bo
rng = np.random.default_rng(77)
c0 = rng.uniform(0.0, 0.79, (150, 120))
n = 200
c = np.ones((n, n))
for i in range(n):
    c[i, i] = 0.30 + 0.001 * rng.uniform()
    if i + 1 < n:
        c[i, i + 1] = 0.10 + 0.001 * rng.uniform()
_, xo, yo = bo.lapjv(c, True, 0.8)
bad = 0
for t in range(150):
    if t % 3 == 0:
        bts.lapjv(c0, True, 0.8)
    _, x, y = bts.lapjv(c, True, 0.8)
    if not np.array_equal(x, xo):
        bad += 1
        d = np.nonzero(x != xo)[0]
        print("trial", t, "diff rows", d[:10], "x", x[d[:10]], "xo", xo[d[:10]])
print("bad", bad, "of 150")
