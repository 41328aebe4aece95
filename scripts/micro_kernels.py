"""HBM-bound standalone kernels against the 8 TB/s roofline (C ABI, device-resident operands, hipEvent timing):
batched Kalman predict / update (1152 / 1184 B per track) and the N x M Mahalanobis gating matrix (8 B per pair)."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
lib = pkg._lib.load(require_gpu=True)
ctx = pkg._lib.default_context().handle
P = pkg._lib.ptr
s = pkg._lib.current_stream()

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps      # us

rng = np.random.default_rng(0)
for n, m in ((4096, 1024), (10000, 2000), (16384, 4096), (262144, 0)):
    mean = torch.from_numpy(rng.normal(500, 100, (n, 8))).cuda()
    A = rng.normal(size=(n, 8, 8)); cov = torch.from_numpy(A @ A.transpose(0, 2, 1) + 10 * np.eye(8)).cuda()
    z32 = torch.from_numpy(rng.normal(500, 100, (n, 4)).astype(np.float32)).cuda()
    t = timed(lambda: pkg._lib.check(lib.bts_kf_multi_predict(ctx, P(mean), P(cov), None, None, n, None, s)))
    print("multi_predict n=%6d  %7.1f us  %6.0f GB/s (%.0f%% of 8 TB/s)" % (n, t, n * 1152 / t / 1e3, n * 1152 / t / 1e3 / 80))
    t = timed(lambda: pkg._lib.check(lib.bts_kf_update(ctx, P(mean), P(cov), None, None, P(z32), None, n, s)))
    print("update        n=%6d  %7.1f us  %6.0f GB/s (%.0f%% of 8 TB/s)" % (n, t, n * 1184 / t / 1e3, n * 1184 / t / 1e3 / 80))
    if m:
        z = torch.from_numpy(rng.normal(500, 100, (m, 4))).cuda()
        out = torch.empty((n, m), dtype=torch.float64, device="cuda")
        t = timed(lambda: pkg._lib.check(lib.bts_kf_gating(ctx, P(mean), P(cov), n, P(z), m, 0, 0, P(out), s)))
        print("gating  %6d x %4d  %7.1f us  %6.0f GB/s (%.0f%% of 8 TB/s)" % (n, m, t, n * m * 8 / t / 1e3, n * m * 8 / t / 1e3 / 80))

# ---- the dense matrix forms of the standalone API at the target size
n, m, D = 10000, 2000, 2048
def boxes(k):
    x, y = rng.uniform(0, 12000, k), rng.uniform(0, 12000, k)
    return torch.from_numpy(np.stack([x, y, x + rng.uniform(30, 80, k), y + rng.uniform(60, 160, k)], 1)).cuda()
a, b = boxes(n), boxes(m)
cost = torch.empty((n, m), dtype=torch.float64, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_iou_distance(ctx, P(a), None, n, P(b), None, m, P(cost), m, s)))
print("iou_distance %d x %d  %7.1f us  %6.0f GB/s written (%.0f%% of 8 TB/s)" % (n, m, t, n * m * 8 / t / 1e3, n * m * 8 / t / 1e3 / 80))
ft = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).cuda(); fd = torch.from_numpy(rng.standard_normal((m, D)).astype(np.float32)).cuda()
sim = torch.empty((n, m), dtype=torch.float32, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_cos_similarity(ctx, P(ft), n, P(fd), m, D, P(sim), m, s)), reps=5)
print("cos_similarity %d x %d x %d  %7.1f us  %6.1f TFLOP/s f32 (%.0f%% of 157 TF)" % (n, m, D, t, 2.0 * n * m * D / t / 1e6, 2.0 * n * m * D / t / 1e6 / 1.573))
out = torch.empty((n, m), dtype=torch.float64, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_fuse_cost_stage1(ctx, P(cost), m, P(sim), None, m, n, m, 0.5, 0.25, P(out), m, s)))
print("fuse_cost_stage1 %d x %d  %7.1f us  %6.0f GB/s moved (%.0f%% of 8 TB/s)" % (n, m, t, n * m * 20 / t / 1e3, n * m * 20 / t / 1e3 / 80))
k = 2000
curr = torch.zeros((n, D), dtype=torch.float32, device="cuda"); smooth = torch.zeros((n, D), dtype=torch.float32, device="cuda")
has = torch.ones((n,), dtype=torch.uint8, device="cuda")
ti = torch.from_numpy(rng.permutation(n)[:k].astype(np.int32)).cuda(); di = torch.arange(k, dtype=torch.int32, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_feature_ema(ctx, P(curr), P(smooth), P(has), P(ti), P(fd), P(di), k, D, 0.9, s)))
print("feature_ema %d rows x %d  %7.1f us  %6.0f GB/s moved (%.0f%% of 8 TB/s)" % (k, D, t, k * D * 16 / t / 1e3, k * D * 16 / t / 1e3 / 80))
t = timed(lambda: pkg._lib.check(lib.bts_feature_renorm(ctx, P(fd), None, k, D, s)))
print("feature_renorm %d rows x %d  %7.1f us  %6.0f GB/s moved (%.0f%% of 8 TB/s)" % (k, D, t, k * D * 8 / t / 1e3, k * D * 8 / t / 1e3 / 80))

# ---- remaining standalone entry points
nn = 262144
mean = torch.from_numpy(rng.normal(500, 100, (nn, 8))).cuda()
A_ = rng.normal(size=(nn, 8, 8)).astype(np.float64); cov = torch.from_numpy(A_ @ A_.transpose(0, 2, 1) + 10 * np.eye(8)).cuda()
pm = torch.empty((nn, 4), dtype=torch.float64, device="cuda"); pc = torch.empty((nn, 16), dtype=torch.float64, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_kf_project(ctx, P(mean), P(cov), None, nn, P(pm), P(pc), s)))
print("project       n=%6d  %7.1f us  %6.0f GB/s (%.0f%% of 8 TB/s)" % (nn, t, nn * (576 + 160) / t / 1e3, nn * 736 / t / 1e3 / 80))
z32 = torch.from_numpy(rng.normal(500, 100, (nn, 4)).astype(np.float32)).cuda()
t = timed(lambda: pkg._lib.check(lib.bts_kf_initiate(ctx, P(z32), nn, P(mean), P(cov), s)))
print("initiate      n=%6d  %7.1f us  %6.0f GB/s (%.0f%% of 8 TB/s)" % (nn, t, nn * 592 / t / 1e3, nn * 592 / t / 1e3 / 80))
na, nb_ = 2000, 8000
a2, b2 = boxes(na), boxes(nb_)
ga = torch.zeros((na,), dtype=torch.int32, device="cuda"); gb = torch.zeros((nb_,), dtype=torch.int32, device="cuda")
fa = torch.zeros((na,), dtype=torch.uint8, device="cuda"); fb = torch.zeros((nb_,), dtype=torch.uint8, device="cuda")
t = timed(lambda: pkg._lib.check(lib.bts_dedupe_pairs(ctx, P(a2), None, P(ga), na, P(b2), None, P(gb), nb_, 0.15, P(fa), P(fb), s)))
print("dedupe_pairs %d x %d  %7.1f us  (%.1f G pair tests/s)" % (na, nb_, t, na * nb_ / t / 1e3))
