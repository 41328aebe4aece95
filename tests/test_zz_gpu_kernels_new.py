"""GPU parity tests of kernels whose current machine code differs from what the round-1 driver run executed: the sparse
component solver `k_lap` (csrc/lap.hip) and the int8 shadow builder / pair test of the appearance pre-filter (`k_i8_rows`,
csrc/feat_math.h).  They sit in a test_zz_ file so that `pytest -x` reaches the tests of the unchanged kernels and the golden
replays (tests/test_gpu_kernels.py, test_gpu_round2.py, test_gpu_tracker.py) first.  Bars as in tests/test_gpu_kernels.py:
assignment indices bit-exact against the golden vectors (g4_lap.npz, recorded from the reference: D:1682-1693) and the oracle.
"""
import importlib

import numpy as np
import pytest

from tests.conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bts():
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    pkg._lib.load(require_gpu=True)
    return pkg


# ------------------------------------------------------------------------------------------ assignment
def test_lapjv_golden(bts):
    g = golden("g4_lap.npz")
    for name in g["names"]:
        name = str(name)
        m, ua, ub = bts.linear_assignment(g[name + "_cost"], float(g[name + "_thresh"]))
        assert np.array_equal(np.asarray(m, dtype=np.int64).reshape(-1, 2), g[name + "_m"]), name
        assert np.array_equal(ua, g[name + "_ua"]) and np.array_equal(ub, g[name + "_ub"]), name
    for shape in ((0, 5), (4, 0), (0, 0)):
        m, ua, ub = bts.linear_assignment(np.zeros(shape), 0.8)
        assert isinstance(ua, tuple) and isinstance(ub, tuple) and m.shape == (0, 2)        # Q14
        assert ua == tuple(range(shape[0])) and ub == tuple(range(shape[1]))


def _objective(c, x, limit):
    n, m = c.shape
    k = int((x >= 0).sum())
    return float(c[np.nonzero(x >= 0)[0], x[x >= 0]].sum()) + limit / 2 * (n + m - 2 * k)


@pytest.mark.parametrize("seed", range(6))
def test_lapjv_random_vs_oracle(bts, seed):
    """tie-free random costs: the optimum is unique, so indices must match the oracle's JV."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(100 + seed)
    for trial in range(8):
        n, m = (int(v) for v in rng.integers(1, 90, 2))
        c = rng.uniform(0, 1, (n, m))
        c[rng.uniform(size=(n, m)) < rng.uniform(0, 0.97)] = 1.0
        th = float(rng.choice([0.5, 0.7, 0.8]))
        _, xo, yo = bo.lapjv(c, True, th)
        _, x, y = bts.lapjv(c, True, th)
        assert np.array_equal(x, xo) and np.array_equal(y, yo), (seed, trial, n, m)


def test_lapjv_dense_block_and_chain(bts):
    """one big connected component (dense feasible block) and a long alternating chain: the
    augmenting paths must re-route earlier matches."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(77)
    c = rng.uniform(0.0, 0.79, (150, 120))                     # everything feasible: one component
    _, xo, yo = bo.lapjv(c, True, 0.8)
    _, x, y = bts.lapjv(c, True, 0.8)
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    n = 200                                                    # chain: row i likes col i, col i+1
    c = np.ones((n, n))
    for i in range(n):
        c[i, i] = 0.30 + 0.001 * rng.uniform()
        if i + 1 < n:
            c[i, i + 1] = 0.10 + 0.001 * rng.uniform()
    _, xo, yo = bo.lapjv(c, True, 0.8)
    _, x, y = bts.lapjv(c, True, 0.8)
    assert abs(_objective(c, x, 0.8) - _objective(c, xo, 0.8)) < 1e-9
    assert np.array_equal(x, xo)


def test_lapjv_tracking_shaped(bts):
    """4096 x 1024 with the sparsity of a tracking frame (a few feasible entries per column)."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(5)
    n, m = 4096, 1024
    c = np.ones((n, m))
    rows = rng.permutation(n)[:m]
    for j in range(m):
        c[rows[j], j] = rng.uniform(0.02, 0.5)
        for r in rng.integers(0, n, 2):
            c[r, j] = min(c[r, j], rng.uniform(0.3, 0.9))
    _, xo, yo = bo.lapjv(c, True, 0.8)
    _, x, y = bts.lapjv(c, True, 0.8)
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    # consistency of the two maps
    for i in np.nonzero(x >= 0)[0]:
        assert y[x[i]] == i and c[i, x[i]] <= 0.8


def test_prefilter_int8_shadows_bound_is_sound(bts):
    """The appearance pre-filter decides which pairs get the exact float32 dot product from int8 shadows of the rows
    (csrc/feat_math.h: per-row scale, MEASURED residual, prefix norms).  Its promise: a pair it excludes -- at any prefix
    checkpoint of the K loop or on the full row -- has an exact dot product below the gate.  Checked here on the device's
    own shadows / records / test function against float64 NumPy, on well-spread rows, heavy-tailed (post-ReLU-like) rows,
    scaled rows, a zero row and a NaN row, with pairs far from the gate, around it (+-0.03) and above it."""
    import ctypes, torch
    lib = bts._lib.load(require_gpu=True)
    ctx = bts._lib.default_context().handle
    rng = np.random.default_rng(12)
    gate = 0.75
    for d in (384, 512, 2048):
        lay = np.zeros(8, dtype=np.int32)
        bts._lib.check(lib.bts_debug_i8_layout(d, lay.ctypes.data))
        chk, kt_all, stage, nmeta = [int(v) for v in lay[:4]], int(lay[4]), int(lay[5]), int(lay[6])
        assert kt_all == d // stage and nmeta == 20 and all(3 <= c <= kt_all for c in chk) and chk == sorted(chk)
        n = 96
        base = rng.standard_normal((n, d))
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        rows = [base]
        # partners at chosen cosines to the first 48 rows
        for cosv in (0.72, 0.745, 0.755, 0.78, 0.95):
            noise = rng.standard_normal((48, d))
            noise -= (noise * base[:48]).sum(1, keepdims=True) * base[:48]
            noise /= np.linalg.norm(noise, axis=1, keepdims=True)
            rows.append(cosv * base[:48] + np.sqrt(1 - cosv ** 2) * noise)
        relu = np.maximum(rng.standard_normal((32, d)), 0.0) ** 3          # heavy-tailed, half zeros
        rows.append(relu / np.linalg.norm(relu, axis=1, keepdims=True))
        rows.append(0.5 * base[:8]); rows.append(3.0 * base[8:16])           # not unit length
        special = np.zeros((2, d)); special[1, 5] = np.nan
        rows.append(special)
        x = np.ascontiguousarray(np.concatenate(rows).astype(np.float32))
        N = len(x)
        dx = torch.from_numpy(x).cuda()
        dq = torch.empty((N, d), dtype=torch.int8, device="cuda")
        dm = torch.empty((N, nmeta), dtype=torch.float32, device="cuda")
        bts._lib.check(lib.bts_debug_i8_rows(ctx, bts._lib.ptr(dx), N, d, gate, bts._lib.ptr(dq), bts._lib.ptr(dm), bts._lib.current_stream()))
        q, meta = dq.cpu().numpy().astype(np.int64), dm.cpu().numpy().astype(np.float64)
        x64 = x.astype(np.float64)
        fin = np.isfinite(x64).all(axis=1)
        mx = np.abs(x64).max(axis=1)
        ok = fin & (mx > 0)
        # the shadow: symmetric int8, ONE scale for the dimension (127 at 6 / sqrt(d)), saturating
        scale = 127.0 * np.sqrt(d) / 6.0
        assert np.abs(q).max() <= 127
        assert np.abs(q[fin] - np.clip(x64[fin] * scale, -127, 127)).max() <= 0.5 + 2e-3
        assert np.allclose(meta[:, 0], 1.0 / scale, rtol=1e-6) and (q[~fin] == 0).all()
        assert (np.abs(x64[fin] * scale) > 127).any()           # (the scaled rows do saturate: the residual must account for it)
        for c in range(1, 5):       # every checkpoint's float4 repeats {1/s, r}
            assert np.array_equal(meta[:, 4 * c], meta[:, 0]) and np.array_equal(meta[:, 4 * c + 1], meta[:, 1])
        # the record: residual bound really bounds, prefix norms are the quantised row's
        qd = q * meta[:, 0:1]
        res = np.linalg.norm(np.where(np.isfinite(x64), x64, 0.0) - qd, axis=1)
        assert (meta[fin, 1] >= res[fin]).all() and (meta[fin, 1] <= res[fin] * 1.001 + 2e-5).all()
        assert np.isinf(meta[~fin, 1]).all()
        a_full = (qd ** 2).sum(1)
        alpha = np.maximum(np.sqrt(a_full), np.sqrt((np.where(np.isfinite(x64), x64, 0.0) ** 2).sum(1)))
        assert (meta[fin, 18] >= alpha[fin] * (1 - 1e-6)).all() and (meta[fin, 18] <= alpha[fin] * 1.0001 + 1e-7).all()
        assert (meta[fin][:, [2, 6, 10, 14]] == 0).all()          # prefix records carry no cross terms
        # the test function on pairs
        ia = np.concatenate([rng.integers(0, N, 4000), np.tile(np.arange(48), 5), np.arange(N)])
        ib = np.concatenate([rng.integers(0, N, 4000), 96 + np.arange(240), np.arange(N)[::-1]]).astype(np.int32)
        ia = ia.astype(np.int32)
        P = len(ia)
        dia, dib = torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()
        dex = torch.empty((P, 5), dtype=torch.uint8, device="cuda")
        ddot = torch.empty((P, 5), dtype=torch.int32, device="cuda")
        bts._lib.check(lib.bts_debug_i8_pairs(ctx, bts._lib.ptr(dq), bts._lib.ptr(dm), bts._lib.ptr(dq), bts._lib.ptr(dm),
                                              bts._lib.ptr(dia), bts._lib.ptr(dib), P, d,
                                              bts._lib.ptr(dex), bts._lib.ptr(ddot), bts._lib.current_stream()))
        ex, dots = dex.cpu().numpy().astype(bool), ddot.cpu().numpy().astype(np.int64)
        exact = (x64[ia] * x64[ib]).sum(1)                       # float64 dot products of the float32 rows
        for c in range(5):
            upto = d if c == 4 else min(chk[c], kt_all) * stage
            assert np.array_equal(dots[:, c], (q[ia, :upto] * q[ib, :upto]).sum(1))
            bad = ex[:, c] & ~(exact < gate - 1e-6)              # SOUNDNESS: excluded => below the gate (NaN: never excluded)
            assert not bad.any(), (d, c, exact[bad][:5])
        # ... and the bound is not vacuous: far pairs are excluded on the full row and already on a prefix;
        # the band around the gate is what the measured residuals say (a few hundredths)
        far = np.isfinite(exact) & (exact < 0.5)
        assert ex[far, 4].mean() > 0.995
        unit_pairs = (ia < 96 + 240) & (ib < 96 + 240)
        if d >= 2048:
            assert ex[far & unit_pairs, 1].mean() > 0.95       # 7/16 of the row decides nearly every far pair of well-spread rows
        near = unit_pairs & np.isfinite(exact) & (exact < gate - 0.06)
        assert ex[near, 4].all()


def test_lapjv_global_memory_form(bts):
    """More than 2048 non-trivial edges: the solver's global-memory form (union-find in LDS, everything else in the
    work area), chains and small dense blocks included; exact against the dense oracle."""
    import ctypes, torch
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(77)
    n, m = 3000, 1800
    c = np.ones((n, m))
    rows = rng.permutation(n)[:m]
    for j in range(m):
        c[rows[j], j] = rng.uniform(0.02, 0.5)
        for r in rng.integers(0, n, 3):
            c[r, j] = min(c[r, j], rng.uniform(0.2, 0.79))
    assert (c <= 0.8).sum() > 5000
    _, xo, yo = bo.lapjv(c, True, 0.8)
    _, x, y = bts.lapjv(c, True, 0.8)
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
