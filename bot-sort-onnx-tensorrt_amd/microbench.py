"""Standalone-kernel measurements against the MI355X rooflines (8 TB/s HBM, 157 TF float32 MFMA): each C-ABI entry
point on device-resident operands, timed with hipEvents on the current stream, algorithmic bytes (or flops) / time.
Used by bench.py (quick set, in its JSON line) and scripts/micro_kernels.py (full table)."""
from __future__ import annotations

import numpy as np

from . import _lib

HBM_GBS, F32_TF = 8000.0, 157.3


def _timed(fn, reps):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps      # us


def _row(name, shape, us, amount, unit, peak):
    rate = amount / us / 1e3 if unit == "GB/s" else amount / us / 1e6
    return {"kernel": name, "shape": shape, "avg_us": round(us, 2), "achieved": round(rate, 1), "unit": unit,
            "peak": peak, "frac": round(rate / peak, 4)}


def run(full: bool = False, seed: int = 0):
    """-> list of {kernel, shape, avg_us, achieved, unit, peak, frac}."""
    import torch
    lib = _lib.load(require_gpu=True)
    ctx = _lib.default_context().handle
    P, s, chk = _lib.ptr, _lib.current_stream(), _lib.check
    rng = np.random.default_rng(seed)
    out = []

    def state(n):
        mean = torch.from_numpy(rng.normal(500, 100, (n, 8))).cuda()
        a = rng.normal(size=(n, 8, 8))
        return mean, torch.from_numpy(a @ a.transpose(0, 2, 1) + 10 * np.eye(8)).cuda()

    def boxes(k):
        x, y = rng.uniform(0, 12000, k), rng.uniform(0, 12000, k)
        return torch.from_numpy(np.stack([x, y, x + rng.uniform(30, 80, k), y + rng.uniform(60, 160, k)], 1)).cuda()

    for n in ((4096, 10000, 16384, 262144) if full else (262144,)):
        mean, cov = state(n)
        z32 = torch.from_numpy(rng.normal(500, 100, (n, 4)).astype(np.float32)).cuda()
        t = _timed(lambda: chk(lib.bts_kf_multi_predict(ctx, P(mean), P(cov), None, None, n, None, s)), 20)
        out.append(_row("kf_multi_predict", "%d tracks" % n, t, n * 1152, "GB/s", HBM_GBS))
        t = _timed(lambda: chk(lib.bts_kf_update(ctx, P(mean), P(cov), None, None, P(z32), None, n, s)), 20)
        out.append(_row("kf_update", "%d tracks" % n, t, n * 1184, "GB/s", HBM_GBS))
        if full and n == 262144:
            pm = torch.empty((n, 4), dtype=torch.float64, device="cuda")
            pc = torch.empty((n, 16), dtype=torch.float64, device="cuda")
            t = _timed(lambda: chk(lib.bts_kf_project(ctx, P(mean), P(cov), None, n, P(pm), P(pc), s)), 20)
            out.append(_row("kf_project", "%d tracks" % n, t, n * 736, "GB/s", HBM_GBS))
            t = _timed(lambda: chk(lib.bts_kf_initiate(ctx, P(z32), n, P(mean), P(cov), s)), 20)
            out.append(_row("kf_initiate", "%d tracks" % n, t, n * 592, "GB/s", HBM_GBS))
    for n, m in (((4096, 1024), (10000, 2000), (16384, 4096)) if full else ((10000, 2000),)):
        mean, cov = state(n)
        z = torch.from_numpy(rng.normal(500, 100, (m, 4))).cuda()
        g = torch.empty((n, m), dtype=torch.float64, device="cuda")
        t = _timed(lambda: chk(lib.bts_kf_gating(ctx, P(mean), P(cov), n, P(z), m, 0, 0, P(g), s)), 20)
        out.append(_row("kf_gating", "%d x %d" % (n, m), t, n * m * 8, "GB/s", HBM_GBS))
    n, m, d = 10000, 2000, 2048
    a, b = boxes(n), boxes(m)
    cost = torch.empty((n, m), dtype=torch.float64, device="cuda")
    t = _timed(lambda: chk(lib.bts_iou_distance(ctx, P(a), None, n, P(b), None, m, P(cost), m, s)), 20)
    out.append(_row("iou_distance", "%d x %d" % (n, m), t, n * m * 8, "GB/s", HBM_GBS))
    ft = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    fd = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
    sim = torch.empty((n, m), dtype=torch.float32, device="cuda")
    t = _timed(lambda: chk(lib.bts_cos_similarity(ctx, P(ft), n, P(fd), m, d, P(sim), m, s)), 5)
    out.append(_row("cos_similarity", "%d x %d x %d" % (n, m, d), t, 2.0 * n * m * d, "TFLOP/s", F32_TF))
    fused = torch.empty((n, m), dtype=torch.float64, device="cuda")
    t = _timed(lambda: chk(lib.bts_fuse_cost_stage1(ctx, P(cost), m, P(sim), None, m, n, m, 0.5, 0.25, P(fused), m, s)), 20)
    out.append(_row("fuse_cost_stage1", "%d x %d" % (n, m), t, n * m * 20, "GB/s", HBM_GBS))
    if full:
        k = 2000
        curr = torch.zeros((n, d), dtype=torch.float32, device="cuda")
        smooth = torch.zeros((n, d), dtype=torch.float32, device="cuda")
        has = torch.ones((n,), dtype=torch.uint8, device="cuda")
        ti = torch.from_numpy(rng.permutation(n)[:k].astype(np.int32)).cuda()
        di = torch.arange(k, dtype=torch.int32, device="cuda")
        t = _timed(lambda: chk(lib.bts_feature_ema(ctx, P(curr), P(smooth), P(has), P(ti), P(fd), P(di), k, d, 0.9, s)), 20)
        out.append(_row("feature_ema", "%d rows x %d" % (k, d), t, k * d * 16, "GB/s", HBM_GBS))
        t = _timed(lambda: chk(lib.bts_feature_renorm(ctx, P(fd), None, k, d, s)), 20)
        out.append(_row("feature_renorm", "%d rows x %d" % (k, d), t, k * d * 8, "GB/s", HBM_GBS))
    return out
