"""(File name: sorts AFTER the GPU suites that were green on hardware in rounds 1-2, so that under `pytest -x` a failure of
code that has never met a GPU -- round 3 had no GPU access -- cannot hide their result.)

Round-3 GPU parity tests: detection rows that do not arrive unit-norm through the int8 pre-filter, the
one-call list export behind the BoTSORT facade, exact ties (see test_zz_gpu_ties.py for the tie families)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")


@pytest.fixture(scope="module")
def bts():
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    pkg._lib.load(require_gpu=True)
    return pkg


@pytest.mark.parametrize("lo,hi", [(0.8, 0.95), (1.05, 1.25), (0.8, 1.25)])
def test_filter_mode_with_non_unit_detection_norms(bts, lo, hi):
    """A track's curr feature is the RENORMALISED detection row x / |x| (D:492-502, Q8) while this frame's similarities are
    taken with the RAW rows.  Detections arrive with norms in [lo, hi]; boxes teleport every frame so that only the
    appearance term can re-associate; the raw similarity |x| cos sits within +-0.02 of the 0.75 gate.  The int8 pre-filter
    (FILTER) must hand the solver exactly the pairs the dense float32 path (DENSE) finds, and both must equal the oracle --
    with the shadow record of a track taken from a non-unit row unadjusted, FILTER drops true matches for norms < 1."""
    import torch
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(int(lo * 1000 + hi * 10))
    K, D = 1024, 256
    proto = rng.standard_normal((K, D))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    a = bts.DeviceTracker(max_tracks=2304, max_dets=K, feat_dim=D)
    b = bts.DeviceTracker(max_tracks=2304, max_dets=K, feat_dim=D, appearance_mode=1)
    ora = bo.TrackerOracle()
    matched = []
    for f in range(5):
        cx = rng.permutation(K) * 7 + 10
        cy = rng.integers(100, 8000, K)
        boxes = np.stack([cx, cy, cx + 5, cy + 9], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, K) if f == 0 else rng.uniform(0.5, 0.85, K)).astype(np.float32)
        nrm = rng.uniform(lo, hi, K)
        c = (0.75 + rng.uniform(3e-5, 0.02, K) * rng.choice([-1.0, 1.0], K)) / nrm           # raw similarity = |x| cos
        c = np.minimum(c, 0.999)
        if f == 0:
            c[:] = 1.0
        noise = rng.standard_normal((K, D))
        noise -= (noise * proto).sum(1, keepdims=True) * proto
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        unit = c[:, None] * proto + np.sqrt(1 - c[:, None] ** 2) * noise
        feats = (nrm[:, None] * unit).astype(np.float32)
        proto = feats.astype(np.float64) / np.linalg.norm(feats.astype(np.float64), axis=1, keepdims=True)
        perm = rng.permutation(K)
        boxes, scores, feats = boxes[perm], scores[perm], np.ascontiguousarray(feats[perm])
        proto = proto[perm]
        fr = (torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda())
        a.step(*fr, torch.from_numpy(feats.copy()).cuda())
        b.step(*fr, torch.from_numpy(feats.copy()).cuda())
        ora.update_arrays(boxes, scores.astype(np.float64), feats.copy())
        for k in (1, 2, 3):
            ra, rb = a.stage_result(k), b.stage_result(k)
            assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["y"], rb["y"]), (f, k)
            assert np.array_equal(ra["m"], ora.trace["m%d" % k]), (f, k)
        matched.append(len(a.stage_result(1)["m"]))
        for which, lst in ((0, ora.tracked), (1, ora.lost)):
            ta, tb, want = a.export(which), b.export(which), bo.TrackerOracle.table(lst)
            for key in INT_KEYS:
                assert np.array_equal(ta[key], tb[key]) and np.array_equal(ta[key], want[key]), (f, which, key)
        assert a.counts() == b.counts() and a.counts()["errors"] == 0
    # about half of the raw similarities sit inside the gate; a track that misses once keeps its old feature and is out of
    # the game (its cosine with the drifted identity is the product of two): the matches halve per frame -- what matters is
    # that the tracks matched at frame 2+ took their shadows from NON-UNIT rows at frame 1+ (the oracle, run on the CPU
    # beforehand: 515 / 276 / 132 / 73 matches for norms 0.8 .. 0.95)
    assert matched[1] > K // 4 and matched[2] > K // 8 and matched[3] > K // 16
    a.close(); b.close()


def test_export_lists_equals_the_per_list_export(bts):
    """bts_tracker_export_lists (both lists, pinned memory, one synchronisation) = two bts_tracker_export calls;
    bts_tracker_read_column = bts_tracker_read_slot(field 6) for every slot."""
    import torch
    synth = bts.synth
    st = synth.SyntheticStream(96, 48, 64, seed=11, churn=0.1)
    trk = bts.DeviceTracker(max_tracks=512, max_dets=96, feat_dim=64, frame_rate=3)
    for f in range(45):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())
        if f % 4 and f < 40:
            continue
        snap = trk.export_lists(f)
        c = trk.counts()
        assert snap.counts == c and snap.n(0) == c["n_tracked"] and snap.n(1) == c["n_lost"]
        for which in (0, 1):
            t = trk.export(which)
            i = snap.ints[which]
            for col, key in enumerate(("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len", "det_index", "slot")):
                assert np.array_equal(i[:, col], t[key]), (f, which, key)
            assert np.array_equal(snap.mean[which], t["mean"]) and np.array_equal(snap.tlbr[which], t["tlbr"])
            assert np.array_equal(snap.score[which], t["score"])
        ids, states = trk.read_column(0), trk.read_column(1)
        for s in list(snap.ints[0][:5, 7]) + list(snap.ints[1][:5, 7]):
            one = trk.read_slot(int(s), 6)
            assert (ids[s], states[s]) == (one[0], one[1])
    assert c["n_lost"] > 0 and c["n_removed"] > 0
    trk.close()


def test_facade_lists_are_lazy_and_match_the_oracle(bts):
    """BoTSORT.update_arrays over the real device tracker: TrackLists equal the oracle's lists; removed tracks reach
    removed_stracks frozen; a mirror kept across frames follows its track (CPU twin: tests/test_facade_host.py)."""
    from oracle import bts_oracle as bo
    st = bts.synth.SyntheticStream(96, 48, 64, seed=12, churn=0.1)
    bot = bts.BoTSORT(max_tracks=1024, max_dets=96, feat_dim=64, frame_rate=3)
    ora = bo.TrackerOracle(frame_rate=3)
    kept = None
    for f in range(50):
        fr = st.next()
        out = bot.update_arrays(fr.boxes, fr.scores, fr.feats.copy())
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        assert isinstance(out, bts.tracker.TrackList) and len(out) == len(ora.tracked) and len(bot.lost_stracks) == len(ora.lost)
        assert [t.track_id for t in bot.removed_stracks] == [t.track_id for t in ora.removed], f
        if f % 5 == 4:
            continue                               # lists never touched on these frames
        assert [t.track_id for t in out] == [t.track_id for t in ora.tracked]
        assert [t.track_id for t in bot.lost_stracks] == [t.track_id for t in ora.lost]
        assert [t.state for t in bot.lost_stracks] == [t.state for t in ora.lost]
        for mine, ref in list(zip(out, ora.tracked))[::5]:
            assert np.abs(mine.mean - np.asarray(ref.mean, dtype=np.float64)).max() <= 1e-6
            assert np.abs(mine.tlbr - np.asarray(ref.tlbr, dtype=np.float64)).max() <= 1e-6
        if f == 3:
            kept = out[0]
        if kept is not None:
            now = [t for t in ora.tracked + ora.lost if t.track_id == kept.track_id]
            if now:
                assert kept.frame_id == now[0].frame_id and kept.state == now[0].state
    live = {t.track_id for t in ora.tracked + ora.lost}
    assert [t.track_id for t in bot.removed_stracks] == [t.track_id for t in ora.removed] and len(bot.removed_stracks) > 0
    for t in bot.removed_stracks:
        if t.track_id not in live:
            assert t.state == bts.TrackState.Removed and t.covariance.shape == (8, 8) and t.face_curr_feature is None


def _frames_on_device(bts, K, M, D, n, **kw):
    import torch
    st = bts.synth.SyntheticStream(K, M, D, seed=99, **kw)
    out = []
    for _ in range(n):
        fr = st.next()
        out.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda()))
    torch.cuda.synchronize()
    return out


def _same_run(bts, frames_a, frames_b, mk_a, mk_b):
    import torch
    a, b = mk_a(), mk_b()
    for f, (fa, fb) in enumerate(zip(frames_a, frames_b)):
        a.step(*fa); b.step(*fb)
        for k in (1, 2, 3):
            ra, rb = a.stage_result(k), b.stage_result(k)
            assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["y"], rb["y"]), (f, k)
        for which in (0, 1):
            ta, tb = a.export(which), b.export(which)
            for key in INT_KEYS:
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)
        assert a.counts() == b.counts() and a.counts()["errors"] == 0
    a.close(); b.close()


@pytest.mark.parametrize("K,M,D", [(512, 256, 256), (4096, 1024, 256)])
def test_opt_in_experiments_change_no_result(bts, monkeypatch, K, M, D):
    """BTS_PREDICT_FIRST (predict workgroups in front of the pre-filter's persistent GEMM workgroups) and BTS_CVT_OVERLAP
    (detection-row conversion of frame t+1 on a side stream under frame t) are scheduling experiments: every assignment,
    list and Kalman mean must be bit-identical to the default schedule.  Frames are resident before the steps are
    enqueued (the overlap mode's contract); each tracker gets its own copy of them (rows are renormalised in place)."""
    n = 14
    mk = lambda: bts.DeviceTracker(max_tracks=2 * K, max_dets=max(K, M), feat_dim=D)      # (churn: the table also holds the lost and the newly born)
    for env in ("BTS_PREDICT_FIRST", "BTS_CVT_OVERLAP"):
        fa, fb = _frames_on_device(bts, K, M, D, n, churn=0.05), _frames_on_device(bts, K, M, D, n, churn=0.05)

        def mk_exp():
            monkeypatch.setenv(env, "1")
            try:
                return mk()
            finally:
                monkeypatch.delenv(env)
        _same_run(bts, fa, fb, mk, mk_exp)


def _crowd_blocks(rng, shapes, limit):
    """Block-diagonal cost matrix: every block a crowd-shaped component (rows = tracks around a few detections, each row
    with 1-4 feasible columns of its block), continuous costs (no ties)."""
    n, m = sum(s[0] for s in shapes), sum(s[1] for s in shapes)
    c = np.ones((n, m))
    r0 = c0 = 0
    for r, cn, scarce in shapes:
        for i in range(r):
            # `scarce`: most rows crowd onto the first few columns, so that some columns lose every row they touch
            pool = cn if not scarce or rng.uniform() < 0.3 else max(1, cn // 4)
            deg = int(rng.integers(1, 5)) if r * 2.5 <= 480 else int(rng.integers(1, 3))      # (<= 512 edges per component)
            for j in rng.choice(pool, size=min(pool, deg), replace=False):
                c[r0 + i, c0 + j] = rng.uniform(0.02, limit - 0.01)
        for j in range(cn):                       # every column touched (a column without edges is not part of a component)
            if (c[r0:r0 + r, c0 + j] < 1.0).sum() == 0:
                c[r0 + int(rng.integers(r)), c0 + j] = rng.uniform(0.02, limit - 0.01)
        if r > 254 or cn > 64:                    # the big blocks are ONE component each: a path through all their columns
            for j in range(cn):
                for i in (j % r, (j + 1) % r):
                    if c[r0 + i, c0 + j] >= 1.0:
                        c[r0 + i, c0 + j] = rng.uniform(0.02, limit - 0.01)
            for i in range(r):                    # ... and every row on it
                if (c[r0 + i, c0:c0 + cn] < 1.0).sum() == 0 or i >= cn:
                    c[r0 + i, c0 + i % cn] = min(c[r0 + i, c0 + i % cn], rng.uniform(0.02, limit - 0.01))
        r0, c0 = r0 + r, c0 + cn
    p, q = rng.permutation(n), rng.permutation(m)      # rows / columns of a component are not contiguous ids
    return c[p][:, q]


@pytest.mark.parametrize("seed", range(4))
def test_lapjv_column_form_components_vs_oracle(bts, seed):
    """Components with more rows than columns (dense crowds) are solved with one shortest-path search per COLUMN
    (solve_tall_cols, lap.hip; round 3, never run on a GPU): 17 ... 254 rows around 3 ... 64 columns, up to 512 edges, with
    columns that must stay free and searches that re-route earlier matches.  Tie-free costs: the optimum is unique, so the
    indices must equal the oracle's dense JV.  Shapes with no more rows than columns (row form) ride along."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(900 + seed)
    shapes = [(17, 4, False), (40, 12, True), (64, 20, False), (65, 64, False), (120, 30, True), (200, 50, False),
              (254, 64, False), (250, 8, True), (30, 40, False), (90, 3, True), (16, 5, False), (33, 33, False)]
    order = rng.permutation(len(shapes))
    limit = float(rng.choice([0.5, 0.8]))
    c = _crowd_blocks(rng, [shapes[k] for k in order], limit)
    _, xo, yo = bo.lapjv(c, True, limit)
    _, x, y = bts.lapjv(c, True, limit)
    assert np.array_equal(x, xo) and np.array_equal(y, yo), seed
    assert (yo < 0).sum() > 0 and (xo < 0).sum() > 0, "the case must leave columns and rows unmatched"


@pytest.mark.parametrize("case", ["lds-prep", "global-prep", "wide", "beyond"])
def test_lapjv_components_beyond_a_wavefronts_lds_vs_oracle(bts, case):
    """Components with more than 254 rows, 64 columns or 512 edges (the core of a dense crowd) are queued and solved by one
    wavefront with the workgroup's whole LDS (solve_big_cols, lap.hip; round 3) instead of the global-memory solver:
    `lds-prep` -- few enough non-trivial edges for the LDS preparation, which now admits such components; `global-prep` --
    more than 4096 non-trivial edges; `wide` -- more columns than rows; `beyond` -- a component past even that solver's
    limits (600 columns) still takes the global-memory path.  Tie-free costs: indices equal the oracle's dense JV."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng({"lds-prep": 1, "global-prep": 2, "wide": 3, "beyond": 4}[case])
    limit = 0.8
    big = {"lds-prep": [(300, 100, False), (260, 70, True)], "global-prep": [(400, 120, True), (270, 66, False)],
           "wide": [(100, 300, False), (200, 200, False)], "beyond": [(700, 600, False), (300, 90, True)]}[case]
    small = [(int(rng.integers(2, 40)), int(rng.integers(1, 12)), False) for _ in range(30 if case != "global-prep" else 420)]
    shapes = big + small
    order = rng.permutation(len(shapes))
    c = _crowd_blocks(rng, [shapes[k] for k in order], limit)
    hard = int((c <= limit).sum())
    assert hard > 4300 if case == "global-prep" else hard < 4000 or case == "beyond", hard      # (which preparation runs)
    _, xo, yo = bo.lapjv(c, True, limit)
    _, x, y = bts.lapjv(c, True, limit)
    assert np.array_equal(x, xo) and np.array_equal(y, yo), case


@pytest.mark.parametrize("case", ["b-crowd", "b-too-many-components", "past-b"])
def test_lapjv_lds_preparation_layouts_vs_oracle(bts, case):
    """The solver's LDS preparation has two layouts since round 3: up to 4096 non-trivial edges with up to 2048 components, or
    up to 6144 edges with up to 1536 components (dense crowds: 5 000+ edges in a few hundred components).  `b-crowd`: ~5 000
    edges in crowd-shaped components (layout B); `b-too-many-components`: ~4 600 edges in 2 300 two-edge components -- layout B
    finds more components than it holds and the global-memory preparation starts over; `past-b`: more than 6144 edges."""
    from oracle import bts_oracle as bo
    rng = np.random.default_rng({"b-crowd": 11, "b-too-many-components": 12, "past-b": 13}[case])
    limit = 0.8
    if case == "b-too-many-components":
        shapes = [(2, 1, False)] * 2300
    else:
        shapes = [(int(rng.integers(20, 64)), int(rng.integers(6, 20)), bool(rng.integers(2))) for _ in range(56 if case == "b-crowd" else 80)]
    c = _crowd_blocks(rng, shapes, limit)
    hard = int((c <= limit).sum())
    assert {"b-crowd": 4300 < hard < 6000, "b-too-many-components": 4300 < hard < 6000, "past-b": hard > 6300}[case], hard
    _, xo, yo = bo.lapjv(c, True, limit)
    _, x, y = bts.lapjv(c, True, limit)
    assert np.array_equal(x, xo) and np.array_equal(y, yo), case
