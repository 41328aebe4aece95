"""(File name: sorts after the suites that have been green on hardware, like the other test_zz_gpu_* files.)

Round-6 GPU parity tests -- written while GPU access was closed and run on the CPU emulation of the kernel sources (tests/hip_emu):
the opt-in throughput form BTS_FILTER_AHEAD (csrc/tracker.hip, enqueue_step): the appearance pre-filter of frame f runs on the side
stream ahead of the step that consumes it -- a MAIN pass over all table slots (valid for tracks whose shadow was not rewritten in
step f-1) and a DET x DET pass frame f-1 x frame f (a track rewritten in step f-1 holds exactly the shadow and track-side record of
the detection it absorbed) -- and the stage-1 tiles pick each pool row's bitmap row through the selector tk_finalize writes.  Bars:
identical assignment indices, ordered lists incl. slot numbers and Kalman means per frame against the default forms AND the oracle
(D:1414-1639), at the sizes VERDICT r5 names (384 x 256 x 2048, 4224 x 2048 x 256 with churn), and on an appearance-only stream
whose identities disappear and come back, so that both kinds of bitmap row decide assignments.  tests/test_emu_gpu_subset.py runs
the appearance-only case under the emulator's adversarial stream schedules with fault injection."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")
AHEAD = {"BTS_FILTER_AHEAD": 1}


@pytest.fixture(scope="module")
def bts():
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    pkg._lib.load(require_gpu=True)
    return pkg


def _dev(fr):
    import torch
    return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())


def _same_frame(a, b, f):
    for k in (1, 2, 3):
        ra, rb = a.stage_result(k), b.stage_result(k)
        assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["y"], rb["y"]), (f, k)
    for which in (0, 1):
        ta, tb = a.export(which), b.export(which)
        for key in INT_KEYS + ("slot",):
            assert np.array_equal(ta[key], tb[key]), (f, which, key)
        assert np.array_equal(ta["mean"], tb["mean"]), (f, which)


class Flicker:
    """K identities with fixed unit prototypes; every frame a random `visible` share of them is detected, at a teleported box (no
    IoU overlap with anything of the previous frame) and with a feature whose cosine to the identity's LAST SHOWN feature lies inside
    the appearance gate.  Only appearance candidates can associate a detection: an identity shown in consecutive frames is found
    through the det x det rows of the bitmap, one that was hidden for a frame or more (a lost track) through the main-pass rows."""

    def __init__(self, K, D, seed, visible=0.7, cos=(0.8, 0.92)):
        self.rng = np.random.default_rng(seed)
        self.K, self.D, self.visible, self.cos = K, D, visible, cos
        p = self.rng.standard_normal((K, D))
        self.last = p / np.linalg.norm(p, axis=1, keepdims=True)
        self.f = 0

    def next(self):
        rng, K, D = self.rng, self.K, self.D
        show = np.ones(K, bool) if self.f == 0 else rng.uniform(size=K) < self.visible
        ids = np.nonzero(show)[0]
        n = len(ids)
        cx = rng.permutation(4 * K)[:n] * 7 + 10
        cy = rng.integers(100, 8000, n)
        boxes = np.stack([cx, cy, cx + 5, cy + 9], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, n) if self.f == 0 else rng.uniform(0.5, 0.85, n)).astype(np.float32)
        c = np.ones(n) if self.f == 0 else rng.uniform(self.cos[0], self.cos[1], n)
        base = self.last[ids]
        noise = rng.standard_normal((n, D))
        noise -= (noise * base).sum(1, keepdims=True) * base
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        feats = (c[:, None] * base + np.sqrt(1 - c[:, None] ** 2) * noise).astype(np.float32)
        self.last[ids] = feats.astype(np.float64) / np.linalg.norm(feats.astype(np.float64), axis=1, keepdims=True)
        perm = rng.permutation(n)

        class Fr:
            pass
        fr = Fr()
        fr.boxes, fr.scores, fr.feats = boxes[perm], scores[perm], np.ascontiguousarray(feats[perm])
        fr.ids = ids[perm]
        self.f += 1
        return fr


def _pair(bts, T, M, D, extra=None, **kw):
    ref = bts.DeviceTracker(max_tracks=T, max_dets=M, feat_dim=D, **kw)
    exp = bts.DeviceTracker(max_tracks=T, max_dets=M, feat_dim=D, forms=dict(AHEAD, **(extra or {})), **kw)
    eff = exp.forms()
    assert eff["filter_ahead"] == "1" and eff["lists_ahead"] == "1" and eff["predict_front"] == "1" and eff["cvt_overlap"] == "1", eff
    assert ref.forms()["filter_ahead"] == "0"
    exp.inputs_resident(True)
    return ref, exp


@pytest.mark.parametrize("extra", ["", "merge_early+renorm_fused+self", "planes", "sorted", "sorted+smooth+merge_early"])
def test_filter_ahead_on_an_appearance_only_stream_with_hidden_identities(bts, extra):
    """160 identities x 128-d, 70 % shown per frame, boxes teleport: every association of frames 1.. is an appearance candidate read
    through pool_cbrow -- det x det rows for identities shown in consecutive frames, main-pass rows for re-found lost tracks.  The
    form = the default forms frame by frame (indices, lists, slots, means), nothing synchronises the host in between except the
    reads of the comparison itself, and both kinds of row carried matches."""
    envs = {"merge_early": "BTS_MERGE_EARLY", "renorm_fused": "BTS_RENORM_FUSED", "self": "BTS_LAP_SELF", "planes": "BTS_SHADOW_PLANES",
            "sorted": "BTS_EDGES_SORTED", "smooth": "BTS_SMOOTH_DEFERRED"}
    K, D = 160, 128
    ref, exp = _pair(bts, K + 96, K, D, extra={envs[k]: 1 for k in extra.split("+") if k})
    assert exp.forms()["edges"] == ("sorted" if "sorted" in extra else "tiles")      # (sorted: candidate LISTS of both side passes, rows translated)
    st = Flicker(K, D, 5)
    refound, cont = 0, 0
    prev_ids = None
    for f in range(14):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        _same_frame(ref, exp, f)
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, exp.counts())
        if f >= 1:
            m = len(exp.stage_result(1)["m"])
            shown_before = np.isin(fr.ids, prev_ids).sum()
            cont += min(m, int(shown_before))
            refound += max(0, m - int(shown_before))
            assert exp.debug_filter()["candidates"] >= m          # every appearance match was a candidate of one of the two passes
        prev_ids = fr.ids
    assert cont > 5 * K // 2 and refound > K // 4, (cont, refound)      # both kinds of bitmap row decided assignments
    ref.close(); exp.close()


@pytest.mark.parametrize("extra", ["", "sorted+smooth"])
def test_filter_ahead_at_384x256x2048_vs_default_and_oracle(bts, extra):
    """VERDICT r5's first size: 384-row table, 256 identities x 2048-d (16 K stages, prefix checkpoints in use), 10 % churn: births,
    expiries, re-found tracks, low-score detections (stage 2 rewrites tracks from LOW-score rows: det x det rows of detections that are
    not in the high-score list of their own frame).  Form = default forms = oracle, frame by frame."""
    from oracle import bts_oracle as bo
    K, M, D = 256, 160, 2048
    ref, exp = _pair(bts, 384, 256, D, extra={"BTS_EDGES_SORTED": 1, "BTS_SMOOTH_DEFERRED": 1} if extra else None)
    ora = bo.TrackerOracle(faithful_iou=False)
    st = bts.synth.SyntheticStream(K, M, D, seed=21, churn=0.1)
    cands = 0
    for f in range(6):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        _same_frame(ref, exp, f)
        for k in (1, 2, 3):
            assert np.array_equal(exp.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        if f >= 1:
            cands += exp.debug_filter()["candidates"]
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert cands > 2 * M
    ref.close(); exp.close()


@pytest.mark.parametrize("extra", ["", "sorted"])
def test_filter_ahead_at_4224x2048x256_with_churn(bts, extra):
    """VERDICT r5's second size: a 4224-row table, 2048 detections x 256-d per frame with 10 % churn: 33 x 16 main-pass tiles and
    16 x 16 det x det tiles over the side passes' persistent workgroups (several tiles each: next-tile prefetch, stop prediction),
    ragged extents on both sides.  Form = default forms frame by frame, and the oracle's assignment on the last frame."""
    from oracle import bts_oracle as bo
    K, M, D = 2048, 2048, 256          # (the stream's first frame shows every identity: K <= max_dets)
    ref, exp = _pair(bts, 4224, 2048, D, extra={"BTS_EDGES_SORTED": 1} if extra else None)
    ora = bo.TrackerOracle(faithful_iou=False)          # (faithful_iou=False: the vectorised IoU form of the checker)
    st = bts.synth.SyntheticStream(K, M, D, seed=9, churn=0.1)
    for f in range(4):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        _same_frame(ref, exp, f)
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, exp.counts())
    for k in (1, 2, 3):
        assert np.array_equal(exp.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), k
    fs = exp.debug_filter()
    assert fs["tiles"] >= 33 * 12 and fs["candidates"] >= len(exp.stage_result(1)["m"]) // 2       # (main pass: every slot row tile x the live column tiles)
    ref.close(); exp.close()


@pytest.mark.parametrize("smooth", [0, 1, 2])
@pytest.mark.parametrize("resident", [True, False])
def test_filter_ahead_without_host_synchronisation_between_steps(bts, resident, smooth):
    """24 frames of the appearance-only stream stepped back to back, digests compared with the default forms' every sixth frame
    only: on a GPU the side passes of frame f really run under step f-1's kernels.  Without the residency promise the side stream is
    ordered behind the caller's stream at every step (correct, no overlap).  On the CPU emulation this test is also run under
    HIPEMU_STREAMS=sidelate / mainlate with fault injection (tests/test_emu_gpu_subset.py).  The final state digest covers the
    covariances and the `curr` / `smooth` feature rows of both lists."""
    tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
    K, D = 96, 128
    # (smooth = 1: + the `smooth` EMA on the third stream; 2: + the sorted edges -- the side stream also clears and fills the
    #  candidate lists -- and merge-early)
    extra = {"BTS_SMOOTH_DEFERRED": 1 if smooth else 0}
    if smooth == 2:
        extra.update(BTS_EDGES_SORTED=1, BTS_MERGE_EARLY=1)
    ref, exp = _pair(bts, K + 64, K, D, extra=extra)
    assert exp.forms()["smooth_deferred"] == str(1 if smooth else 0) and exp.forms()["edges"] == ("sorted" if smooth == 2 else "tiles")
    exp.inputs_resident(resident)
    st = Flicker(K, D, 31, visible=0.75)
    frames = [_dev(st.next()) for _ in range(24)]
    frames2 = [(b, s, f.clone()) for (b, s, f) in frames]
    import torch
    torch.cuda.synchronize()
    for f, (fa, fb) in enumerate(zip(frames, frames2)):
        ref.step(*fa)
        exp.step(*fb)
        if f % 6 == 5:
            assert tune._digest(ref) == tune._digest(exp), f
    assert tune._state_digest(ref) == tune._state_digest(exp)
    assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert ref.counts()["n_tracked"] > K // 2           # (the stream sustained itself on appearance alone)
    ref.close(); exp.close()


def test_filter_ahead_is_ruled_out_where_it_does_not_apply(bts):
    """The form needs the pre-filter, body features only, the tile or sorted consumers and eager launches: elsewhere it is not in effect
    (bts_tracker_forms reports forms IN EFFECT) and the tracker behaves like the default."""
    for kw in (dict(appearance_mode=1), dict(face_dim=128), dict(use_graph=1), dict(record_costs=True)):
        t = bts.DeviceTracker(max_tracks=256, max_dets=64, feat_dim=128, forms=AHEAD, **kw)
        eff = t.forms()
        t.close()
        assert eff["filter_ahead"] == "0", (kw, eff)
    t = bts.DeviceTracker(max_tracks=256, max_dets=64, feat_dim=128, forms=dict(AHEAD, BTS_EDGES_GRID=1))      # (the neighbour search reads the bitmap by pool row)
    eff = t.forms()
    t.close()
    assert eff["filter_ahead"] == "1" and eff["edges"] == "tiles", eff


def test_filter_ahead_edge_frames_reset_and_gmc(bts):
    """Empty frames (the side passes have no tiles: nothing may be launched with an empty grid), a frame of low-score detections only
    (Q19 shape: no stage-1 columns), detection counts that change from frame to frame (the det x det pass has the PREVIOUS frame's row
    count), a camera-motion warp between steps and a reset in mid-stream (the det x det rows of the frame before the reset are stale:
    nothing may consult them).  Form = oracle = default forms, frame by frame."""
    import torch
    from oracle import bts_oracle as bo
    rng = np.random.default_rng(4)
    D = 128
    ref, exp = _pair(bts, 64, 16, D)
    ora = bo.TrackerOracle()
    protos = rng.standard_normal((8, D))
    protos /= np.linalg.norm(protos, axis=1, keepdims=True)

    def frame(boxes, scores, ids, n):
        m = len(boxes)
        f = (protos[ids] + 0.05 * rng.standard_normal((m, D))).astype(np.float32) if m else np.zeros((0, D), np.float32)
        if m:
            f /= np.linalg.norm(f, axis=1, keepdims=True)
        b = np.asarray(boxes, dtype=np.int32).reshape(m, 4)
        s = np.asarray(scores, dtype=np.float32)
        for t in (ref, exp):
            t.step(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(f.copy()).cuda())
        ora.update_arrays(b, s.astype(np.float64), f.copy())
        _same_frame(ref, exp, n)
        for k in (1, 2, 3):
            assert np.array_equal(exp.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (n, k)
        for which, lst in ((0, ora.tracked), (1, ora.lost)):
            want = bo.TrackerOracle.table(lst)
            for key in INT_KEYS:
                assert np.array_equal(exp.export(which)[key], want[key]), (n, key)
        assert exp.counts()["errors"] == 0

    frames = [
        ([], [], []),
        ([[10, 10, 60, 120], [200, 50, 260, 190], [400, 400, 450, 500]], [0.95, 0.97, 0.99], [0, 1, 2]),
        ([], [], []),
        ([[11, 11, 61, 121], [201, 52, 262, 191]], [0.3, 0.35], [0, 1]),                       # only low scores
        ([[900, 900, 950, 1000]], [0.6], [2]),                                                 # teleported: appearance only (main-pass row)
        ([[12, 12, 62, 122], [203, 52, 263, 192], [905, 903, 955, 1003], [600, 10, 640, 90]], [0.96, 0.93, 0.7, 0.99], [0, 1, 2, 3]),
        ([[13, 12, 63, 122], [1500, 700, 1550, 800]], [0.96, 0.8], [0, 2]),                    # id 2 teleports again right after a match (det x det row)
        ([[14, 13, 64, 123], [204, 53, 264, 193], [1502, 701, 1552, 801], [601, 11, 641, 91], [50, 700, 90, 790]], [0.9, 0.5, 0.6, 0.7, 0.95], [0, 1, 2, 3, 4]),
    ]
    for n, (b, s, ids) in enumerate(frames):
        frame(b, s, ids, n)
    H = np.asarray([[1.01, -0.02, 4.0], [0.02, 1.01, -2.5]])
    for t in (ref, exp):
        t.gmc(H)
    for lst in (ora.tracked, ora.lost):          # (STrack.multi_gmc on the oracle's lists, like tests/test_gpu_tracker.py::test_tracker_gmc)
        if lst:
            m, c = bo.KalmanOracle.multi_gmc(np.asarray([t.mean for t in lst], dtype=np.float64), np.asarray([t.cov for t in lst], dtype=np.float64), H)
            for t, mm, cc in zip(lst, m, c):
                t.mean, t.cov = mm, cc
    frame([[18, 10, 69, 121], [1521, 735, 1572, 836]], [0.9, 0.8], [0, 2], 100)
    for t in (ref, exp):
        t.reset()
    ora = bo.TrackerOracle()
    for n, (b, s, ids) in enumerate(frames[1:6]):
        frame(b, s, ids, 200 + n)
    ref.close(); exp.close()


def test_smooth_deferred_rows_equal_the_default_forms_every_frame(bts):
    """BTS_SMOOTH_DEFERRED (with BTS_FILTER_AHEAD): a step writes a rewritten track's `curr` row only; tk_smooth, on a third stream
    behind the step, recomputes `smooth = normalise(alpha smooth + beta curr)` (newborn: smooth = curr) -- ema_row's own arithmetic on
    the row it stored.  After EVERY frame of a stream with births, re-found tracks and low-score matches the exported covariances,
    `curr` and `smooth` rows of both lists are bit-identical to the default forms' (the export is ordered behind tk_smooth), as are
    indices, lists and means; also across a reset."""
    K, M, D = 96, 64, 256
    ref, exp = _pair(bts, 192, 96, D, extra={"BTS_SMOOTH_DEFERRED": 1, "BTS_MERGE_EARLY": 1})
    assert exp.forms()["smooth_deferred"] == "1" and ref.forms()["smooth_deferred"] == "0"
    st = bts.synth.SyntheticStream(K, M, D, seed=17, churn=0.15)
    for f in range(12):
        if f == 8:
            ref.reset(); exp.reset()
            st = bts.synth.SyntheticStream(K, M, D, seed=18, churn=0.15)
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        _same_frame(ref, exp, f)
        for which in (0, 1):
            a, b = ref.export_state(which, cov=True, body=True, face=False), exp.export_state(which, cov=True, body=True, face=False)
            for key in ("covariance", "body_curr_feature", "body_smooth_feature"):
                assert a[key].shape == b[key].shape and np.array_equal(a[key], b[key]), (f, which, key)
        assert exp.counts()["errors"] == 0
    t = bts.DeviceTracker(max_tracks=64, max_dets=32, feat_dim=128, forms={"BTS_SMOOTH_DEFERRED": 1})       # needs the ahead mode's streams
    assert t.forms()["smooth_deferred"] == "0"
    t.close()
    ref.close(); exp.close()


def test_filter_ahead_sorted_edges_when_the_candidate_slots_overflow(bts):
    """All identities share one appearance (pairwise cosines ~ 0.97): every (track, detection) pair is a candidate, a 64 x 64 quadrant
    of a side pass holds thousands -- far more than a list slot -- so the candidate wavefronts of the sorted form SWEEP the quadrant's
    bitmap words instead, rows translated from slot / previous-detection space one row at a time, in both passes.  Boxes move
    slowly, so IoU decides among the many candidates.  Form = default forms = oracle."""
    from oracle import bts_oracle as bo
    K, D = 150, 128
    ref, exp = _pair(bts, 256, 160, D, extra={"BTS_EDGES_SORTED": 1})
    assert exp.forms()["edges"] == "sorted"
    ora = bo.TrackerOracle(faithful_iou=False)
    rng = np.random.default_rng(12)
    proto = rng.standard_normal(D)
    proto /= np.linalg.norm(proto)
    cx, cy = rng.permutation(K) * 40 + 20, rng.integers(50, 3000, K)
    swept = 0
    for f in range(6):
        show = np.ones(K, bool) if f == 0 else rng.uniform(size=K) < 0.8
        ids = np.nonzero(show)[0]
        cx = cx + rng.integers(-2, 3, K)
        boxes = np.stack([cx[ids], cy[ids], cx[ids] + 30, cy[ids] + 60], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, len(ids)) if f == 0 else rng.uniform(0.5, 0.85, len(ids))).astype(np.float32)
        feats = proto[None, :] + 0.02 * rng.standard_normal((len(ids), D))
        feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)

        class Fr:
            pass
        fr = Fr()
        fr.boxes, fr.scores, fr.feats = boxes, scores, np.ascontiguousarray(feats)
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        _same_frame(ref, exp, f)
        for k in (1, 2, 3):
            assert np.array_equal(exp.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        assert exp.counts()["errors"] == 0
        if f >= 1:
            swept += exp.debug_filter()["candidates"]
    assert swept > 5 * 64 * 64          # (quadrants far beyond a slot's capacity)
    ref.close(); exp.close()


@pytest.mark.parametrize("mode", ["motion_gate", "tie_mode", "config5", "motion_gate+sorted"])
def test_the_ahead_forms_under_the_trackers_other_modes(bts, mode):
    """The consumers the ahead mode touches exist in several instantiations: the GATE forms of the stage-1 tiles (motion_gate), the
    exact-tie step behind every assignment (tie_mode 1: its cost source is the stage's boxes and edges), config 5's arithmetic
    (float32 Kalman rows, binary16 costs).  BTS_FILTER_AHEAD + BTS_SMOOTH_DEFERRED + merge-early (+ the sorted edges where the mode
    allows them: the sorted form is ruled out WITH the motion gate, and then the tiles must take over) against the default forms under
    the SAME mode, frame by frame: indices, lists, slots, means; at the end covariances and feature rows."""
    tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
    kw = {"motion_gate": dict(motion_gate=1), "tie_mode": dict(tie_mode=1), "config5": dict(kalman_dtype=1, cost_dtype=2),
          "motion_gate+sorted": dict(motion_gate=1)}[mode]
    extra = {"BTS_SMOOTH_DEFERRED": 1, "BTS_MERGE_EARLY": 1}
    if mode in ("tie_mode", "motion_gate+sorted"):
        extra["BTS_EDGES_SORTED"] = 1
    K, M, D = 120, 80, 128
    ref, exp = _pair(bts, 256, 128, D, extra=extra, **kw)
    eff = exp.forms()
    assert eff["smooth_deferred"] == "1" and eff["edges"] == ("sorted" if mode == "tie_mode" else "tiles"), eff
    st = bts.synth.SyntheticStream(K, M, D, seed=23, churn=0.15)
    for f in range(8):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        _same_frame(ref, exp, f)
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, exp.counts())
    assert tune._state_digest(ref) == tune._state_digest(exp)
    ref.close(); exp.close()
