"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes) and checked
against the golden vectors recorded from the reference (tests/golden/*.npz) and the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for IoU costs, fused costs and assignment indices;
Kalman state within 1e-4 -- held here to 1e-9 relative (the 4x4 Cholesky runs in float64 like
SciPy's, differing only in LAPACK's operation order); float32 ReID similarities within 2e-6
(float32 summation order).
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


def close(a, b, rel=1e-9, abs_=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b)
    tol = abs_ + rel * np.maximum(np.abs(a), np.abs(b))
    assert (err <= tol).all(), float((err - tol).max())


# ------------------------------------------------------------------------------------------ Kalman
def test_kf_initiate_bitwise(bts):
    g = golden("g1_kalman.npz")
    kf = bts.KalmanFilter()
    for i in range(len(g["meas32"])):
        m, c = kf.initiate(g["meas32"][i])
        assert m.dtype == np.float32 and c.dtype == np.float32
        assert np.array_equal(m, g["init_mean"][i]) and np.array_equal(c, g["init_cov"][i])


def test_kf_predict_bitwise(bts):
    g = golden("g1_kalman.npz")
    kf = bts.KalmanFilter()
    sm, sc = g["state_mean"], g["state_cov"]
    m, c = kf.multi_predict(sm.copy(), sc.copy())
    assert np.array_equal(m, g["multi_predict_mean"]) and np.array_equal(c, g["multi_predict_cov"])
    m1, c1 = kf.predict(sm[3], sc[3])
    assert np.array_equal(m1, g["predict_mean"][3]) and np.array_equal(c1, g["predict_cov"][3])
    mz, cz = kf.multi_predict(sm.copy(), sc.copy(), zero_velocity=g["not_tracked"])
    assert np.array_equal(mz, g["multi_predict_z_mean"]) and np.array_equal(cz, g["multi_predict_z_cov"])


def test_kf_predict_float32_batch(bts):
    """frame-2 case: a pool whose rows are all float32 (fresh from initiate): NumPy evaluates the
    process noise in float32 (quirk Q10)."""
    from oracle import bts_oracle as bo
    g = golden("g1_kalman.npz")
    want_m, want_c = bo.KalmanOracle().multi_predict(g["init_mean"].copy(), g["init_cov"].copy())
    m, c = bts.KalmanFilter().multi_predict(g["init_mean"].copy(), g["init_cov"].copy())
    assert np.array_equal(m, want_m) and np.array_equal(c, want_c)


def test_kf_project_bitwise(bts):
    g = golden("g1_kalman.npz")
    kf = bts.KalmanFilter()
    for i in (0, 7, 31):
        z, s = kf.project(g["state_mean"][i], g["state_cov"][i])
        assert np.array_equal(z, g["project_mean"][i]) and np.array_equal(s, g["project_cov"][i])


def test_kf_update(bts):
    g = golden("g1_kalman.npz")
    kf = bts.KalmanFilter()
    m, c = kf.update_batch(g["multi_predict_mean"], g["multi_predict_cov"], g["update_z"])
    close(m, g["update_mean"])
    close(c, g["update_cov"], rel=1e-9, abs_=1e-9)
    um, uc = kf.update(g["multi_predict_mean"][5], g["multi_predict_cov"][5], g["update_z"][5])
    assert np.array_equal(um, m[5]) and np.array_equal(uc, c[5])


def test_kf_update_float32_state(bts):
    """unconfirmed tracks are updated straight from their float32 initiate() state (Q5/Q10)."""
    from oracle import bts_oracle as bo
    g = golden("g1_kalman.npz")
    ko = bo.KalmanOracle()
    kf = bts.KalmanFilter()
    for i in range(6):
        m0, c0 = g["init_mean"][i], g["init_cov"][i]
        z = (m0[:4] + np.float32(1.5)).astype(np.float32)
        wm, wc = ko.update(m0, c0, z)
        gm, gc = kf.update(m0, c0, z)
        close(gm, wm)
        close(gc, wc, abs_=1e-9)


def test_kf_gating(bts):
    g = golden("g1_kalman.npz")
    kf = bts.KalmanFilter()
    for i in range(4):
        for op in (False, True):
            for metric in ("maha", "gaussian"):
                d = kf.gating_distance(g["state_mean"][i], g["state_cov"][i], g["gate_meas"], only_position=op,
                                       metric=metric)
                close(d, g["gate_%d_%d_%s" % (i, int(op), metric)], rel=1e-10)
    # batched form: all tracks x all measurements
    d = kf.gating_distance(g["state_mean"][:4], g["state_cov"][:4], g["gate_meas"])
    assert d.shape == (4, 16)
    close(d[2], g["gate_2_0_maha"], rel=1e-10)
    with pytest.raises(ValueError):
        kf.gating_distance(g["state_mean"][0], g["state_cov"][0], g["gate_meas"], metric="nope")


def test_kf_multi_gmc(bts):
    g, g1 = golden("g7_gmc.npz"), golden("g1_kalman.npz")
    m, c = bts.KalmanFilter().multi_gmc(g1["state_mean"], g1["state_cov"], g["H"])
    close(m, g["mean"], rel=1e-13, abs_=1e-12)
    close(c, g["cov"], rel=1e-12, abs_=1e-12)


# ------------------------------------------------------------------------------------------ IoU
@pytest.mark.parametrize("tag", ["f64xf32", "f32xf32", "f64xf64", "f32xf64", "f64xg32", "f32xg32"])
def test_iou_distance_golden(bts, tag):
    g = golden("g2_iou.npz")
    a, b, want = g[tag + "_a"], g[tag + "_b"], g[tag + "_cost"]
    got = bts.iou_distance(list(a), list(b))
    assert got.shape == want.shape
    if str(g[tag + "_dtype"]) == "float32":
        assert np.abs(got - want.astype(np.float64)).max() <= 6e-8     # `1 - iou` evaluated in f64 here
    else:
        assert np.array_equal(got, want)


def test_iou_distance_empty_and_large(bts):
    from oracle import bts_oracle as bo
    g = golden("g2_iou.npz")
    assert bts.iou_distance([], list(g["f64xf32_b"])).shape == (0, 32)
    assert bts.iou_distance(list(g["f64xf32_a"]), []).shape == (64, 0)
    rng = np.random.default_rng(3)
    n, m = 700, 530                       # several tiles, ragged edges
    xa, ya = rng.uniform(0, 3000, n), rng.uniform(0, 3000, n)
    a = np.stack([xa, ya, xa + rng.uniform(30, 80, n), ya + rng.uniform(60, 160, n)], 1)
    xb, yb = np.rint(rng.uniform(0, 3000, m)), np.rint(rng.uniform(0, 3000, m))
    b = np.stack([xb, yb, xb + np.rint(rng.uniform(30, 80, m)), yb + np.rint(rng.uniform(60, 160, m))], 1).astype(np.float32)
    b[:40] = np.rint(a[:40]).astype(np.float32)       # guaranteed overlaps
    want = bo.iou_cost(list(a), list(b))
    got = bts.iou_distance(list(a), list(b))
    assert np.array_equal(got, want)
    assert (got < 1).sum() >= 40


# ------------------------------------------------------------------------------------------ fused costs
def test_fused_costs_golden(bts):
    g = golden("g3_fused.npz")
    for k in range(int(g["n_rec"])):
        got = bts.fuse_cost_stage1(g["s1_%d_iou" % k], g["s1_%d_body_sim" % k], g["s1_%d_face_sim" % k])
        assert np.array_equal(got, g["s1_%d_cost" % k])
        if "s3_%d_cost" % k in g.files:
            sim = bts.cosine_similarity(g["s3_%d_trk_feats" % k], g["s3_%d_det_feats" % k])
            got3 = bts.fuse_cost_stage3(g["s3_%d_iou" % k], sim)
            assert np.abs(got3 - g["s3_%d_cost" % k]).max() <= 2e-6


def test_face_fixup_and_thresholds(bts):
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(9)
    iou_d = rng.uniform(0, 1, (40, 33))
    body = rng.uniform(0, 1, (40, 33)).astype(np.float32)
    face = rng.uniform(0, 1, (40, 33)).astype(np.float32)
    body[0, :4] = [0.75, np.float32(0.75) + np.float32(1e-7), 0.7499999, 1.0]     # exactly at 0.25 distance
    face[1, :3] = [0.9999999, np.float32(1.0), 0.99999994]                        # the "0.9999999" rule
    iou_d[2, :3] = [0.5, 0.5000000001, 0.4999999]
    want = bo.fuse_cost_stage1(iou_d, body.copy(), bo.face_similarity_fixup(face))
    assert np.array_equal(bts.fuse_cost_stage1(iou_d, body, face), want)
    want3 = bo.fuse_cost_stage3(iou_d, body.copy())
    assert np.array_equal(bts.fuse_cost_stage3(iou_d, body), want3)
    want_nf = bo.fuse_cost_stage1(iou_d, body.copy(), np.zeros_like(face))
    assert np.array_equal(bts.fuse_cost_stage1(iou_d, body, None), want_nf)


# ------------------------------------------------------------------------------------------ cosine
@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("shape", [(5, 3, 64), (130, 257, 2048), (300, 129, 384), (64, 64, 100), (10, 7, 30)])
def test_cosine_similarity(bts, which, shape):
    n, m, d = shape
    rng = np.random.default_rng(n + m)
    t = rng.standard_normal((n, d)).astype(np.float32)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = rng.standard_normal((m, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[: min(n, m)] = t[: min(n, m)] * 0.9 + q[: min(n, m)] * 0.1        # some large similarities
    ctx = bts._lib.default_context()
    ctx.set_cos_kernel(which)
    try:
        got = bts.cosine_similarity(t, q)
    finally:
        ctx.set_cos_kernel(0)
    want = np.maximum(0.0, t.astype(np.float64) @ q.astype(np.float64).T)
    assert got.dtype == np.float32 and got.shape == (n, m)
    assert np.abs(got - want).max() <= 2e-6
    # asymmetric operands: a transposed result would fail (n != m or t != q)
    assert np.abs(got[0] - want[0]).max() <= 2e-6 and np.abs(got[:, 0] - want[:, 0]).max() <= 2e-6


# ------------------------------------------------------------------------------------------ features / dedupe
def test_feature_ema_and_renorm(bts):
    import torch
    from oracle import bts_oracle as bo
    lib = bts._lib.load(require_gpu=True)
    ctx = bts._lib.default_context()
    rng = np.random.default_rng(4)
    d, t, k = 2048, 6, 4
    dets = rng.standard_normal((5, d)).astype(np.float32)
    dets /= np.linalg.norm(dets, axis=1, keepdims=True)
    recs = []
    for i in range(t):
        f = rng.standard_normal(d).astype(np.float32)
        f /= np.linalg.norm(f)
        recs.append(bo.TrackRec([0, 0, 1, 1], 1.0, f.copy()))
    curr = torch.from_numpy(np.stack([r.body_curr for r in recs])).cuda()
    smooth = torch.from_numpy(np.stack([r.body_smooth for r in recs])).cuda()
    has = torch.ones(t, dtype=torch.uint8).cuda()
    has[5] = 0
    trk = torch.tensor([4, 0, 2, 5], dtype=torch.int32).cuda()
    det = torch.tensor([1, 3, 0, 2], dtype=torch.int32).cuda()
    dd = torch.from_numpy(dets).cuda()
    bts._lib.check(lib.bts_feature_ema(ctx.handle, curr.data_ptr(), smooth.data_ptr(), has.data_ptr(), trk.data_ptr(),
                                       dd.data_ptr(), det.data_ptr(), k, d, 0.9, bts._lib.current_stream()))
    got_c, got_s = curr.cpu().numpy(), smooth.cpu().numpy()
    for ti, di in ((4, 1), (0, 3), (2, 0)):
        recs[ti].push_body(dets[di].copy())
        assert np.array_equal(got_c[ti], dets[di])
        assert np.abs(got_s[ti] - recs[ti].body_smooth).max() <= 2e-7
    fresh = dets[2] / np.linalg.norm(dets[2])
    assert np.abs(got_s[5] - fresh).max() <= 2e-7 and has.cpu().numpy()[5] == 1
    assert np.array_equal(got_c[1], recs[1].body_curr)                       # untouched rows
    # in-place renormalisation of detection rows (Q8)
    raw = (rng.standard_normal((3, d)) * 3).astype(np.float32)
    dr = torch.from_numpy(raw).cuda()
    idx = torch.tensor([2, 0], dtype=torch.int32).cuda()
    bts._lib.check(lib.bts_feature_renorm(ctx.handle, dr.data_ptr(), idx.data_ptr(), 2, d, bts._lib.current_stream()))
    out = dr.cpu().numpy()
    for r in (0, 2):
        assert np.abs(out[r] - raw[r] / np.linalg.norm(raw[r])).max() <= 2e-7
    assert np.array_equal(out[1], raw[1])


def test_dedupe_pairs(bts):
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(8)
    n, m = 300, 270
    xa, ya = rng.uniform(0, 1500, n), rng.uniform(0, 1500, n)
    a = np.stack([xa, ya, xa + rng.uniform(30, 80, n), ya + rng.uniform(60, 160, n)], 1)
    b = a[rng.integers(0, n, m)] + rng.normal(0, 1.0, (m, 4))
    b[100:] += 500
    age_a = rng.integers(0, 50, n)
    age_b = rng.integers(0, 50, m)
    pd = bo.iou_cost(list(a), list(b))
    want_a, want_b = np.zeros(n, dtype=bool), np.zeros(m, dtype=bool)
    for p, q in zip(*np.where(pd < 0.15)):
        if age_a[p] > age_b[q]:
            want_b[q] = True
        else:
            want_a[p] = True
    from importlib import import_module
    da, db = import_module("bot-sort-onnx-tensorrt_amd.matching").dedupe_flags(list(a), list(b), age_a, age_b, 0.15)
    assert want_a.sum() > 0 and want_b.sum() > 0
    assert np.array_equal(da, want_a) and np.array_equal(db, want_b)


# ------------------------------------------------------------------------------------------ part association
def test_associate_parts_golden(bts):
    g = golden("g8_parts.npz")
    for tag, picks in (("head", 1), ("hand", 2)):
        idx, used = bts.associate_parts(g[tag + "_base"], g[tag + "_base_c"], g[tag + "_target"], g[tag + "_target_c"], picks=picks)
        assert np.array_equal(idx, g[tag + "_idx"]) and np.array_equal(used, g[tag + "_used"])


@pytest.mark.parametrize("nb,nt,picks,span", [(1, 1, 1, 50), (7, 0, 1, 50), (64, 300, 2, 400), (30, 500, 3, 60), (300, 120, 1, 900)])
def test_associate_parts_vs_oracle(bts, nb, nt, picks, span):
    """Random crowds, incl. a base with > 64 overlapping targets (span 60: everything overlaps), targets
    taken beforehand and exact IoU / distance ties."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(nb * 1000 + nt)

    def boxes(n, lo, hi):
        xy = rng.integers(0, span, (n, 2)); wh = rng.integers(lo, hi, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(np.int64)

    bb, tb = boxes(nb, 30, 120), boxes(nt, 8, 40)
    if nt > 20:
        tb[:10] = tb[10:20]
    bc, tc = (bb[:, :2] + bb[:, 2:]) // 2, (tb[:, :2] + tb[:, 2:]) // 2
    used0 = rng.random(nt) < 0.1
    want, wused = bo.associate_parts(bb, bc, tb, tc, used=used0, picks=picks)
    idx, used = bts.associate_parts(bb, bc, tb, tc, used=used0, picks=picks)
    assert np.array_equal(idx, want) and np.array_equal(used, wused)


def test_find_most_relevant_object_and_attach(bts):
    """Reference-shaped calls: one-object query (D:1763-1791) and the whole D:1303-1411 block."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(5)
    dets = []
    for cls, n, lo, hi in ((0, 12, 60, 160), (1, 14, 15, 40), (2, 20, 10, 30), (3, 9, 8, 25)):
        for _ in range(n):
            x, y = (int(v) for v in rng.integers(0, 300, 2)); w, h = (int(v) for v in rng.integers(lo, hi, 2))
            dets.append(bts.Box(classid=cls, score=0.9, x1=x, y1=y, x2=x + w, y2=y + h, cx=(2 * x + w) // 2, cy=(2 * y + h) // 2))
    bodies, heads, hands, faces = bts.attach_parts(dets)
    arr = lambda L: (np.asarray([[o.x1, o.y1, o.x2, o.y2] for o in L]), np.asarray([[o.cx, o.cy] for o in L]))
    for bases, targets, picks, get in ((heads, faces, 1, lambda b: [b.face]), (bodies, heads, 1, lambda b: [b.head]),
                                       (bodies, hands, 2, lambda b: [b.hand1, b.hand2])):
        want, wused = bo.associate_parts(*arr(bases), *arr(targets), picks=picks)
        for b, row in zip(bases, want):
            assert [targets[k] if k >= 0 else None for k in row] == get(b)
        assert [t.is_used for t in targets] == list(wused)
    assert any(b.head is not None and b.head.face is not None for b in bodies)
    # single query, twice: the second call must skip the part taken by the first
    body, pool = bodies[0], [bts.Hand.of(h) for h in hands]
    first = bts.find_most_relevant_object(base_obj=body, target_objs=pool)
    second = bts.find_most_relevant_object(base_obj=body, target_objs=pool)
    assert first is body.hand1 or (first is not None and (first.x1, first.y1) == (body.hand1.x1, body.hand1.y1))
    assert second is None or second is not first
    assert bts.find_most_relevant_object(base_obj=body, target_objs=[]) is None


def test_bbox_iou_by_box_vs_oracle(bts):
    """D:1715-1729 on detector records, positional construction as the reference's callers do it (D:1303-1411)."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(9)
    for _ in range(40):
        x, y, u, v = (int(t) for t in rng.integers(0, 120, 4)); w, h, p, q = (int(t) for t in rng.integers(5, 90, 4))
        a = bts.Body(0, 0, 0.9, x, y, x + w, y + h, x + w // 2, y + h // 2, False, None, None, None)
        b = bts.Head(0, 1, 0.8, u, v, u + p, v + q, u + p // 2, v + q // 2, False)
        got = bts.bbox_iou_by_box(a, b)
        want = bo.box_iou_int([a.x1, a.y1, a.x2, a.y2], [b.x1, b.y1, b.x2, b.y2])
        assert abs(got - want) <= 1e-12, (got, want)
        if want == 0.0:
            assert got == 0.0


def test_errors_are_loud(bts):
    lib = bts._lib.load(require_gpu=True)
    ctx = bts._lib.default_context()
    rc = lib.bts_iou_distance(ctx.handle, None, None, 4, None, None, 4, None, 4, None)
    assert rc == -1 and b"null" in lib.bts_last_error()
    with pytest.raises(bts.BtsError):
        bts._lib.check(rc)
    rc = lib.bts_lapjv(ctx.handle, None, 2, 2, 2, float("inf"), None, None, None)
    assert rc == -1
