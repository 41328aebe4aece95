"""Standalone-kernel measurements against the MI355X rooflines (8 TB/s HBM, 157 TF float32 MFMA): each C-ABI entry
point on device-resident operands, timed with hipEvents on the current stream, algorithmic bytes (or flops) / time.
Used by bench.py (quick set, in its JSON line) and scripts/micro_kernels.py (full table)."""
from __future__ import annotations

import numpy as np

from . import _lib

HBM_GBS, F32_TF = 8000.0, 157.3


ROTATE_BYTES = 800e6      # every kernel cycles through operand sets totalling more than this: three times the 256 MB
                          # Infinity Cache, so no launch finds its operands cached by an earlier one
HBM_COPY_GBS = 6300.0     # the guide's achievable streaming rate: a figure above it means a cache served the reads


def _nsets(bytes_per_set):
    return max(3, int(-(-ROTATE_BYTES // max(1.0, float(bytes_per_set)))))


def _timed(fn, reps, nsets):
    """fn(k) launches on operand set k; `reps` launches rotate through the nsets sets (warm-up touches each set once)."""
    import torch
    for k in range(min(nsets, 4)):
        fn(k)
    torch.cuda.synchronize()
    reps = max(reps, nsets)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(reps):
        fn(k % nsets)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps      # us


def _row(name, shape, us, amount, unit, peak, nsets=1, set_bytes=0):
    rate = amount / us / 1e3 if unit == "GB/s" else amount / us / 1e6
    row = {"kernel": name, "shape": shape, "avg_us": round(us, 2), "achieved": round(rate, 1), "unit": unit,
           "peak": peak, "frac": round(rate / peak, 4), "operand_sets": nsets, "rotated_MB": round(nsets * set_bytes / 1e6)}
    if unit == "GB/s" and rate > HBM_COPY_GBS:
        row["note"] = "above the achievable HBM copy rate: cache-resident, not an HBM figure"
    return row


def run(full: bool = False, seed: int = 0):
    """-> list of {kernel, shape, avg_us, achieved, unit, peak, frac, operand_sets, rotated_MB}."""
    import torch
    lib = _lib.load(require_gpu=True)
    ctx = _lib.default_context().handle
    P, s, chk = _lib.ptr, _lib.current_stream(), _lib.check
    rng = np.random.default_rng(seed)
    out = []

    def copies(t, k):
        return [t] + [t.clone() for _ in range(k - 1)]

    def state(n):
        mean = torch.from_numpy(rng.normal(500, 100, (n, 8))).cuda()
        a = rng.normal(size=(min(n, 4096), 8, 8))
        c = a @ a.transpose(0, 2, 1) + 10 * np.eye(8)
        return mean, torch.from_numpy(np.tile(c, (-(-n // len(c)), 1, 1))[:n].copy()).cuda()

    def boxes(k):
        x, y = rng.uniform(0, 12000, k), rng.uniform(0, 12000, k)
        return torch.from_numpy(np.stack([x, y, x + rng.uniform(30, 80, k), y + rng.uniform(60, 160, k)], 1)).cuda()

    for n in ((4096, 10000, 16384, 262144) if full else (262144,)):
        mean, cov = state(n)
        k = _nsets(n * 576)
        means, covs = copies(mean, k), copies(cov, k)
        z32 = torch.from_numpy(rng.normal(500, 100, (n, 4)).astype(np.float32)).cuda()
        t = _timed(lambda i: chk(lib.bts_kf_multi_predict(ctx, P(means[i]), P(covs[i]), None, None, n, None, s)), 20, k)
        out.append(_row("kf_multi_predict", "%d tracks" % n, t, n * 1152, "GB/s", HBM_GBS, k, n * 576))
        t = _timed(lambda i: chk(lib.bts_kf_update(ctx, P(means[i]), P(covs[i]), None, None, P(z32), None, n, s)), 20, k)
        out.append(_row("kf_update", "%d tracks" % n, t, n * 1184, "GB/s", HBM_GBS, k, n * 576))
        if full and n == 262144:
            pms = [torch.empty((n, 4), dtype=torch.float64, device="cuda") for _ in range(k)]
            pcs = [torch.empty((n, 16), dtype=torch.float64, device="cuda") for _ in range(k)]
            t = _timed(lambda i: chk(lib.bts_kf_project(ctx, P(means[i]), P(covs[i]), None, n, P(pms[i]), P(pcs[i]), s)), 20, k)
            out.append(_row("kf_project", "%d tracks" % n, t, n * 736, "GB/s", HBM_GBS, k, n * 736))
            t = _timed(lambda i: chk(lib.bts_kf_initiate(ctx, P(z32), n, P(means[i]), P(covs[i]), s)), 20, k)
            out.append(_row("kf_initiate", "%d tracks" % n, t, n * 592, "GB/s", HBM_GBS, k, n * 576))
        del means, covs
    for n, m in (((4096, 1024), (10000, 2000), (16384, 4096)) if full else ((10000, 2000),)):
        mean, cov = state(n)
        z = torch.from_numpy(rng.normal(500, 100, (m, 4))).cuda()
        k = _nsets(n * m * 8)
        gs = [torch.empty((n, m), dtype=torch.float64, device="cuda") for _ in range(k)]
        t = _timed(lambda i: chk(lib.bts_kf_gating(ctx, P(mean), P(cov), n, P(z), m, 0, 0, P(gs[i]), s)), 20, k)
        out.append(_row("kf_gating", "%d x %d" % (n, m), t, n * m * 8, "GB/s", HBM_GBS, k, n * m * 8))
        del gs
    n, m, d = 10000, 2000, 2048
    a, b = boxes(n), boxes(m)
    k = _nsets(n * m * 8)
    costs = [torch.empty((n, m), dtype=torch.float64, device="cuda") for _ in range(k)]
    t = _timed(lambda i: chk(lib.bts_iou_distance(ctx, P(a), None, n, P(b), None, m, P(costs[i]), m, s)), 20, k)
    out.append(_row("iou_distance", "%d x %d" % (n, m), t, n * m * 8, "GB/s", HBM_GBS, k, n * m * 8))
    ft = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    fd = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
    ks = _nsets((n + m) * d * 4 + n * m * 4)
    fts, fds = copies(ft, ks), copies(fd, ks)
    sims = [torch.empty((n, m), dtype=torch.float32, device="cuda") for _ in range(ks)]
    t = _timed(lambda i: chk(lib.bts_cos_similarity(ctx, P(fts[i]), n, P(fds[i]), m, d, P(sims[i]), m, s)), 5, ks)
    out.append(_row("cos_similarity", "%d x %d x %d" % (n, m, d), t, 2.0 * n * m * d, "TFLOP/s", F32_TF, ks, (n + m) * d * 4 + n * m * 4))
    del fts
    kf = min(k, ks)
    fused = [torch.empty((n, m), dtype=torch.float64, device="cuda") for _ in range(kf)]
    t = _timed(lambda i: chk(lib.bts_fuse_cost_stage1(ctx, P(costs[i]), m, P(sims[i]), None, m, n, m, 0.5, 0.25, P(fused[i]), m, s)), 20, kf)
    out.append(_row("fuse_cost_stage1", "%d x %d" % (n, m), t, n * m * 20, "GB/s", HBM_GBS, kf, n * m * 20))
    del costs, sims, fused
    if full:
        # every track row is touched once per pass over the operand sets (k = n rows per launch: the whole table)
        k = n
        ke = _nsets(n * d * 8 + m * d * 4)
        currs = [torch.randn((n, d), dtype=torch.float32, device="cuda") for _ in range(ke)]
        smooths = [torch.randn((n, d), dtype=torch.float32, device="cuda") for _ in range(ke)]
        fdn = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
        has = torch.ones((n,), dtype=torch.uint8, device="cuda")
        ti = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
        di = torch.arange(k, dtype=torch.int32, device="cuda")
        t = _timed(lambda i: chk(lib.bts_feature_ema(ctx, P(currs[i]), P(smooths[i]), P(has), P(ti), P(fdn), P(di), k, d, 0.9, s)), 20, ke)
        out.append(_row("feature_ema", "%d rows x %d" % (k, d), t, k * d * 16, "GB/s", HBM_GBS, ke, n * d * 8))
        del smooths
        t = _timed(lambda i: chk(lib.bts_feature_renorm(ctx, P(currs[i]), None, n, d, s)), 20, ke)
        out.append(_row("feature_renorm", "%d rows x %d" % (n, d), t, n * d * 8, "GB/s", HBM_GBS, ke, n * d * 4))
    return out
