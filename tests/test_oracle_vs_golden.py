"""Pin the CPU oracle (oracle/bts_oracle.py + oracle/lapjv_oracle.c) against the golden vectors
recorded from the reference itself (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import bts_oracle as bo
from tests.conftest import golden


# ------------------------------------------------------------------------------------------ G1
def test_g1_kalman_bitwise():
    g = golden("g1_kalman.npz")
    kf = bo.KalmanOracle()
    n = len(g["meas32"])
    for i in range(n):
        m, c = kf.initiate(g["meas32"][i])
        assert m.dtype == g["init_mean"].dtype and np.array_equal(m, g["init_mean"][i])
        assert np.array_equal(c, g["init_cov"][i])
    sm, sc = g["state_mean"], g["state_cov"]
    for i in range(n):
        m, c = kf.predict(sm[i], sc[i])
        assert np.array_equal(m, g["predict_mean"][i]) and np.array_equal(c, g["predict_cov"][i])
        z, s = kf.project(sm[i], sc[i])
        assert np.array_equal(z, g["project_mean"][i]) and np.array_equal(s, g["project_cov"][i])
    m, c = kf.multi_predict(sm.copy(), sc.copy())
    assert np.array_equal(m, g["multi_predict_mean"]) and np.array_equal(c, g["multi_predict_cov"])
    zm = sm.copy()
    zm[g["not_tracked"], 6:8] = 0
    mz, cz = kf.multi_predict(zm, sc.copy())
    assert np.array_equal(mz, g["multi_predict_z_mean"]) and np.array_equal(cz, g["multi_predict_z_cov"])
    for i in range(n):
        um, uc = kf.update(m[i], c[i], g["update_z"][i])
        assert np.array_equal(um, g["update_mean"][i]) and np.array_equal(uc, g["update_cov"][i])
    for i in range(4):
        for op in (False, True):
            for metric in ("maha", "gaussian"):
                d = kf.gating_distance(sm[i], sc[i], g["gate_meas"], only_position=op, metric=metric)
                assert np.array_equal(d, g["gate_%d_%d_%s" % (i, int(op), metric)])
    assert bo.CHI2INV95[4] == 9.4877


def test_g8_part_association():
    g = golden("g8_parts.npz")
    for tag, picks in (("head", 1), ("hand", 2)):
        idx, used = bo.associate_parts(g[tag + "_base"], g[tag + "_base_c"], g[tag + "_target"], g[tag + "_target_c"], picks=picks)
        assert np.array_equal(idx, g[tag + "_idx"]) and np.array_equal(used, g[tag + "_used"])
        assert (idx >= 0).sum() > 10


def test_g7_multi_gmc():
    g, g1 = golden("g7_gmc.npz"), golden("g1_kalman.npz")
    m, c = bo.KalmanOracle.multi_gmc(g1["state_mean"], g1["state_cov"], g["H"])
    assert np.array_equal(m, g["mean"]) and np.array_equal(c, g["cov"])


# ------------------------------------------------------------------------------------------ G2
@pytest.mark.parametrize("tag", ["f64xf32", "f32xf32", "f64xf64", "f32xf64", "f64xg32", "f32xg32"])
def test_g2_iou(tag):
    g = golden("g2_iou.npz")
    a, b, want = g[tag + "_a"], g[tag + "_b"], g[tag + "_cost"]
    got_pair = bo.iou_cost(list(a), list(b), faithful=True)
    assert got_pair.dtype == want.dtype and np.array_equal(got_pair, want)
    got = bo.iou_cost(list(a), list(b))
    if want.dtype == np.float32:
        # all-float32 matrix: the reference evaluates `1 - iou` in float32 (documented deviation)
        assert np.abs(got - want.astype(np.float64)).max() <= 6e-8
    else:
        assert np.array_equal(got, want)


def test_g2_iou_empty():
    g = golden("g2_iou.npz")
    b = g["f64xf32_b"]
    a = g["f64xf32_a"]
    for faithful in (False, True):
        r = bo.iou_cost([], list(b), faithful=faithful)
        assert r.shape == g["empty_n0_cost"].shape == (0, len(b))
        r = bo.iou_cost(list(a), [], faithful=faithful)
        assert r.shape == g["empty_m0_cost"].shape == (len(a), 0)


# ------------------------------------------------------------------------------------------ G3
def test_g3_fused_costs():
    g = golden("g3_fused.npz")
    n = int(g["n_rec"])
    assert n >= 3
    seen3 = 0
    for k in range(n):
        face = bo.face_similarity_fixup(g["s1_%d_face_sim" % k])
        got = bo.fuse_cost_stage1(g["s1_%d_iou" % k], g["s1_%d_body_sim" % k].copy(), face)
        assert np.array_equal(got, g["s1_%d_cost" % k])
        if "s3_%d_cost" % k in g.files:
            sim = bo.cosine_similarity(g["s3_%d_trk_feats" % k], g["s3_%d_det_feats" % k])
            got3 = bo.fuse_cost_stage3(g["s3_%d_iou" % k], sim)
            assert np.array_equal(np.asarray(got3, dtype=np.float64), g["s3_%d_cost" % k])
            seen3 += 1
    assert seen3 >= 1


# ------------------------------------------------------------------------------------------ G4
def test_g4_linear_assignment():
    g = golden("g4_lap.npz")
    for name in g["names"]:
        name = str(name)
        m, ua, ub = bo.linear_assignment(g[name + "_cost"], float(g[name + "_thresh"]))
        assert np.array_equal(np.asarray(m, dtype=np.int64).reshape(-1, 2), g[name + "_m"])
        assert np.array_equal(ua, g[name + "_ua"]) and np.array_equal(ub, g[name + "_ub"])
        x = np.full(g[name + "_cost"].shape[0], -1)
        for i, j in g[name + "_m"]:
            x[i] = j
        assert np.array_equal(x, g[name + "_scipy_x"])      # independent optimality pin
    for shape in ((0, 5), (4, 0), (0, 0)):
        m, ua, ub = bo.linear_assignment(np.zeros(shape), 0.8)
        tag = "empty_%d_%d" % shape
        assert isinstance(ua, tuple) and isinstance(ub, tuple) and m.shape == (0, 2)   # Q14
        assert np.array_equal(np.asarray(ua, dtype=np.int64), g[tag + "_ua"])
        assert np.array_equal(np.asarray(ub, dtype=np.int64), g[tag + "_ub"])


def test_lapjv_matches_scipy_on_random_tie_free():
    import scipy.optimize
    rng = np.random.default_rng(5)
    for trial in range(40):
        n, m = rng.integers(1, 40, 2)
        c = rng.uniform(0, 1, (n, m))
        c[rng.uniform(size=(n, m)) < rng.uniform(0, 0.9)] = 1.0
        th = float(rng.choice([0.5, 0.7, 0.8]))
        _, x, y = bo.lapjv(c, extend_cost=True, cost_limit=th)
        ext = np.full((n + m, n + m), th / 2)
        ext[n:, m:] = 0
        ext[:n, :m] = c
        r, cc = scipy.optimize.linear_sum_assignment(ext)
        xs = np.full(n, -1)
        for i, j in zip(r, cc):
            if i < n and j < m:
                xs[i] = j
        assert np.array_equal(x, xs)
        for i in range(n):
            if x[i] >= 0:
                assert y[x[i]] == i and c[i, x[i]] <= th


# ------------------------------------------------------------------------------------ G5 / G6
def _replay(name, synth):
    g = golden(name)
    K, M, D, frames = int(g["K"]), int(g["M"]), int(g["D"]), int(g["frames"])
    st = synth.SyntheticStream(K, M, D, seed=int(g["seed"]), churn=float(g["churn"]),
                               face_dim=int(g["face_dim"]))
    trk = bo.TrackerOracle(frame_rate=int(g["frame_rate"]))
    h = hashlib.sha256()
    for f in range(frames):
        fr = st.next()
        h.update(fr.boxes.tobytes()); h.update(fr.scores.tobytes()); h.update(fr.feats.tobytes())
        if fr.face_feats is not None:
            h.update(fr.face_feats.tobytes())
        trk.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats, fr.face_feats)
        for k in (1, 2, 3):
            p = "f%d_s%d_" % (f, k)
            assert np.array_equal(trk.trace["m%d" % k], g[p + "m"]), (f, k)
            assert np.array_equal(trk.trace["u_trk%d" % k], g[p + "u_trk"]), (f, k)
            assert np.array_equal(trk.trace["u_det%d" % k], g[p + "u_det"]), (f, k)
            c = np.asarray(trk.trace["cost%d" % k], dtype=np.float64)
            assert tuple(g[p + "cost_shape"]) == c.shape
            if p + "cost" in g.files:
                assert np.abs(c - g[p + "cost"]).max(initial=0.0) <= 6e-8
                if k != 3:
                    assert np.array_equal(c, g[p + "cost"])
        for nm, lst in (("trk", trk.tracked), ("lost", trk.lost)):
            t = bo.TrackerOracle.table(lst)
            p = "f%d_%s_" % (f, nm)
            for key, v in t.items():
                assert np.array_equal(v, g[p + key]), (f, nm, key)     # means bit-identical too
        assert len(trk.removed) == int(g["f%d_n_removed" % f])
    assert h.hexdigest() == str(g["input_sha256"])      # generator reproduces the recorded inputs
    assert trk._count == int(g["last_id"])


def test_g5_c1_trace(synth):
    _replay("g5_c1_trace.npz", synth)


def test_g6_lifecycle(synth):
    _replay("g6_lifecycle.npz", synth)


def test_g6_faces(synth):
    _replay("g6_faces.npz", synth)


def _replay_big(name, synth, faithful_iou):
    """G9 / G10 hold what make_golden.py --big recorded from the REFERENCE at 4096 x 1024 and 10 000 x 2 000: assignment
    indices, ordered id / state tables, checksums of the means (and the means themselves where they were kept)."""
    g = golden(name)
    K, M, D, frames = int(g["K"]), int(g["M"]), int(g["D"]), int(g["frames"])
    st = synth.SyntheticStream(K, M, D, seed=int(g["seed"]))
    trk = bo.TrackerOracle(faithful_iou=faithful_iou)
    h = hashlib.sha256()
    for f in range(frames):
        fr = st.next()
        h.update(fr.boxes.tobytes()); h.update(fr.scores.tobytes()); h.update(fr.feats.tobytes())
        trk.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            p = "f%d_s%d_" % (f, k)
            assert np.array_equal(np.asarray(trk.trace["m%d" % k]).reshape(-1, 2), g[p + "m"].reshape(-1, 2)), (f, k)
            assert np.array_equal(trk.trace["u_trk%d" % k], g[p + "u_trk"]), (f, k)
            assert np.array_equal(trk.trace["u_det%d" % k], g[p + "u_det"]), (f, k)
        for nm, lst in (("trk", trk.tracked), ("lost", trk.lost)):
            t = bo.TrackerOracle.table(lst)
            p = "f%d_%s_" % (f, nm)
            for key in ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len"):
                assert np.array_equal(t[key], g[p + key]), (f, nm, key)
            want = g[p + "mean_sum"]
            got = np.asarray([t["mean"].sum(), np.abs(t["mean"]).sum()])
            assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), (f, nm)
            if (p + "mean") in g.files and len(t["mean"]):
                assert np.abs(t["mean"] - g[p + "mean"]).max() <= 1e-9, (f, nm)
        assert len(trk.removed) == int(g["f%d_n_removed" % f])
    assert h.hexdigest() == str(g["input_sha256"])
    assert trk._count == int(g["last_id"])


def test_g9_vectorised_oracle_vs_the_reference_at_c3_size(synth):
    """The GPU tests and bench.py's id_match at the big sizes check the device against the VECTORISED form of the oracle
    (faithful_iou=False: same arithmetic, array IoU instead of the per-pair Python loop).  This pins that form to the
    reference itself at 4096 x 1024 x 2048."""
    _replay_big("g9_c3_trace.npz", synth, faithful_iou=False)


def test_g10_vectorised_oracle_vs_the_reference_at_the_headline_size(synth):
    """Same at 10 000 x 2 000 x 2048 (four frames; the dense JV runs on the 12 000 x 12 000 extended matrix)."""
    _replay_big("g10_target_trace.npz", synth, faithful_iou=False)


def test_motion_gate_extension_of_the_oracle(synth):
    """The optional motion gate (not in the reference's loop) is built from pieces the golden vectors pin
    (gating_distance: G1; chi2inv95[4] = 9.4877, D:139-149).  Off it changes nothing; on, a detection teleported 300 px
    keeps its appearance match in the plain cascade and loses it in the gated one."""
    from oracle import bts_oracle as bo
    assert bo.CHI2INV95[4] == 9.4877
    st = synth.SyntheticStream(48, 40, 64, seed=11)
    plain, gated = bo.TrackerOracle(), bo.TrackerOracle(motion_gate=True)
    for f in range(6):
        fr = st.next()
        boxes = fr.boxes.copy()
        if f == 5:
            boxes[3] += 300
        for o in (plain, gated):
            o.update_arrays(boxes, fr.scores.astype(np.float64), fr.feats.copy())
        if f < 5:
            for k in ("m1", "m2", "m3"):
                assert np.array_equal(plain.trace[k], gated.trace[k])
    hi = list(plain.trace["hi_idx"])
    assert 3 in hi
    col = hi.index(3)
    assert col in set(plain.trace["m1"][:, 1]) and col not in set(gated.trace["m1"][:, 1])
    assert gated._count == plain._count + (1 if fr.scores[3] >= 0.9 else 0)
