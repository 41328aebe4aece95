"""Host logic of the BoTSORT facade (bot-sort-onnx-tensorrt_amd/tracker.py) on the CPU: the lazy TrackLists, mirrors kept
across frames, the bulk detection of removed / dropped tracks.  The device tracker is replaced by a TEST DOUBLE that serves
`export_lists / read_column / read_slot / counts` from the CPU oracle (tests may use the oracle; the product never does);
the GPU tests (tests/test_gpu_round2.py) run the same facade over the real device tracker."""
import importlib
import os

import numpy as np
import pytest

from oracle import bts_oracle as bo

pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
trk = pkg.tracker


class OracleBackedTracker:
    """Stands in for DeviceTracker: every track gets a table slot for life (like the device's slot table)."""

    def __init__(self, max_tracks, max_dets, feat_dim, face_dim=0, frame_rate=30, **kw):
        self.cfg = type("Cfg", (), dict(max_tracks=max_tracks, max_dets=max_dets, feat_dim=feat_dim, face_dim=face_dim))()
        self.ora = bo.TrackerOracle(frame_rate=frame_rate)
        self.slot_of, self.rec_of_slot, self.reads = {}, {}, 0

    def step(self, boxes, scores, feats, face_feats=None):
        self.ora.update_arrays(np.asarray(boxes), np.asarray(scores, dtype=np.float64), np.asarray(feats).copy(),
                               None if face_feats is None else np.asarray(face_feats).copy())
        for t in self.ora.tracked + self.ora.lost:
            if t.track_id not in self.slot_of:
                s = len(self.slot_of)
                self.slot_of[t.track_id] = s
                self.rec_of_slot[s] = t

    def _counts(self):
        o = self.ora
        return np.asarray([o.frame_id, len(o.tracked), len(o.lost), len(o.removed), o._count, 0, 0, 0, 0, 0, 0, 0], dtype=np.int32)

    def counts(self):
        return dict(zip(trk.COUNT_KEYS, (int(v) for v in self._counts())))

    def export_lists(self, epoch=0):
        cols = [[], [], [], []]
        for lst in (self.ora.tracked, self.ora.lost):
            n = len(lst)
            ints = np.asarray([[t.track_id, t.state, int(t.is_activated), t.frame_id, t.start_frame, t.tracklet_len, t.det_index,
                                self.slot_of[t.track_id]] for t in lst], dtype=np.int32).reshape(n, 8)
            mean = np.asarray([np.asarray(t.mean, dtype=np.float64) for t in lst]).reshape(n, 8)
            tlbr = np.asarray([np.asarray(t.tlbr, dtype=np.float64) for t in lst]).reshape(n, 4)
            score = np.asarray([t.score for t in lst], dtype=np.float32)
            for c, a in zip(cols, (ints, mean, tlbr, score)):
                c.append(a)
        return trk.Snapshot(epoch, self._counts(), *cols)

    def read_column(self, field):
        out = np.zeros(self.cfg.max_tracks, dtype=np.int32)
        for s, t in self.rec_of_slot.items():
            out[s] = (t.track_id, t.state, t.frame_id)[field]
        return out

    def read_slot(self, slot, field):
        self.reads += 1
        t = self.rec_of_slot[slot]
        return {1: np.asarray(t.cov, dtype=np.float64), 2: t.body_curr, 3: t.body_smooth, 4: t.face_curr, 5: t.face_smooth}[field].copy()


@pytest.fixture
def bot(monkeypatch):
    monkeypatch.setattr(trk, "DeviceTracker", OracleBackedTracker)

    def make(**kw):
        b = pkg.BoTSORT(None, None, None, **kw)

        def update_arrays(boxes, scores, feats, face_feats=None, bodies=None, export=True):      # (the real one moves the arrays to the GPU)
            b.device_tracker.step(boxes, scores, feats, face_feats)
            b.frame_id += 1
            return b._export(bodies) if export else None
        b.update_arrays = update_arrays
        return b
    return make


def _stream(K=48, M=24, D=32, seed=3, churn=0.15):
    return pkg.synth.SyntheticStream(K, M, D, seed=seed, churn=churn)


def test_lazy_lists_mirror_the_oracle_and_build_nothing_until_touched(bot):
    b = bot(frame_rate=3, max_tracks=2048, max_dets=64, feat_dim=32)       # frame_rate 3: tracks expire after 30 frames
    st = _stream()
    ora = b.device_tracker.ora
    kept, touched = {}, set()
    for f in range(70):
        fr = st.next()
        out = b.update_arrays(fr.boxes, fr.scores, fr.feats)
        assert out is b.tracked_stracks and isinstance(out, trk.TrackList) and len(out) == len(ora.tracked)
        assert len(b.lost_stracks) == len(ora.lost)
        assert [t.track_id for t in b.removed_stracks] == [t.track_id for t in ora.removed], f
        if f % 7 == 3:
            assert b._mirror.keys() <= touched, "mirrors were built although nobody touched the lists"
            continue                                  # a frame whose lists the caller never looks at
        for lst, want in ((b.tracked_stracks, ora.tracked), (b.lost_stracks, ora.lost)):
            assert [t.track_id for t in lst] == [t.track_id for t in want]
            touched |= {t.track_id for t in want}
            assert [t.state for t in lst] == [t.state for t in want]
            assert [t.is_activated for t in lst] == [bool(t.is_activated) for t in want]
            assert [(t.frame_id, t.start_frame, t.tracklet_len) for t in lst] == [(t.frame_id, t.start_frame, t.tracklet_len) for t in want]
            for mine, ref in zip(lst, want):
                assert np.array_equal(mine.mean, np.asarray(ref.mean, dtype=np.float64))
                assert np.array_equal(mine.tlbr, np.asarray(ref.tlbr, dtype=np.float64)) and mine.end_frame == ref.frame_id
        if f == 5:
            kept = {t.track_id: t for t in list(b.tracked_stracks)[:6]}
        # an STrack the caller kept is THE object of that track on later frames, and shows the current state when read
        for tid, obj in kept.items():
            now = [t for t in ora.tracked + ora.lost if t.track_id == tid]
            if now:
                assert obj.frame_id == now[0].frame_id and obj.state == now[0].state
                assert np.array_equal(obj.mean, np.asarray(now[0].mean, dtype=np.float64))
                where = b.tracked_stracks if now[0] in ora.tracked else b.lost_stracks
                assert any(t is obj for t in where)
    assert len(ora.removed) > 0, "the stream must exercise removals"
    # removed_stracks = the reference's list (D:1635), entry for entry, in its order: appended in the frame that marks the track
    # removed -- an expired lost track while it is still on lost_stracks (Q6) -- and frozen with its last values and arrays
    # once it has left the lists
    assert [t.track_id for t in b.removed_stracks] == [t.track_id for t in ora.removed]
    live = {t.track_id for t in ora.tracked + ora.lost}
    for t, ref in zip(b.removed_stracks, ora.removed):
        assert t.state == ref.state
        if t.track_id not in live:
            assert t.state == pkg.TrackState.Removed and t.covariance is not None and t.covariance.shape == (8, 8)
            assert t.body_curr_feature is not None and t.face_curr_feature is None
            assert np.array_equal(t.mean, np.asarray(ref.mean, dtype=np.float64))


def test_slices_negative_indices_and_list_helpers(bot):
    b = bot(max_tracks=512, max_dets=64, feat_dim=32)
    st = _stream(churn=0.0)
    for _ in range(3):
        fr = st.next()
        out = b.update_arrays(fr.boxes, fr.scores, fr.feats)
    n = len(out)
    assert n > 8 and out[-1] is out[n - 1] and out[2:5] == [out[2], out[3], out[4]] and out[0] in out
    with pytest.raises(IndexError):
        out[n]
    assert len(out + b.lost_stracks) == n + len(b.lost_stracks) and isinstance(out + [], list)
    assert [t.track_id for t in pkg.tracker.joint_stracks(out, b.lost_stracks)] == [t.track_id for t in out] + [t.track_id for t in b.lost_stracks]
    assert pkg.tracker.sub_stracks(out, out[:3]) == list(out[3:])
    assert repr(out).startswith("TrackList(")


def test_eager_lists_option_and_lazy_arrays(bot):
    b = bot(max_tracks=512, max_dets=64, feat_dim=32, lazy_lists=False)
    st = _stream(churn=0.0)
    fr = st.next()
    out = b.update_arrays(fr.boxes, fr.scores, fr.feats)
    assert isinstance(out, list) and len(b._mirror) == len(out) + len(b.lost_stracks)
    t0 = out[0]
    before = b.device_tracker.reads
    c1, c2 = t0.covariance, t0.covariance
    assert b.device_tracker.reads == before + 1 and c1 is c2                   # fetched once per frame, on access
    assert t0.face_curr_feature is None and t0.face_smooth_feature is None     # face_dim = 0: never asks the device (reference: None)
    assert b.device_tracker.reads == before + 1
    fr = st.next()
    b.update_arrays(fr.boxes, fr.scores, fr.feats)
    assert t0.covariance is not c1 and b.device_tracker.reads == before + 2    # a new frame: re-read on the next access


def test_bodies_are_attached_and_ids_propagated(bot):
    b = bot(max_tracks=512, max_dets=64, feat_dim=32)
    st = _stream(churn=0.0)
    for _ in range(2):
        fr = st.next()
        bodies = [pkg.Body(score=float(s), x1=int(x[0]), y1=int(x[1]), x2=int(x[2]), y2=int(x[3]), head=pkg.Head()) for x, s in zip(fr.boxes, fr.scores)]
        out = b.update_arrays(fr.boxes, fr.scores, fr.feats, bodies=bodies)
    hit = 0
    for t in out:
        if t.frame_id == b.frame_id and t.det_index >= 0:
            assert t.body is bodies[t.det_index] and t.body.trackid == t.track_id and t.body.head.trackid == t.track_id
            hit += 1
    assert hit > 10


def test_zombie_refound_after_removal_like_the_reference(bot):
    """Quirk Q6: a lost track that expires is appended to removed_stracks but stays on lost_stracks for one more frame (the list
    is filtered against the PREVIOUS removals); matched in that frame it is tracked again while still sitting on
    removed_stracks.  The facade's lists must show exactly that (same object on both lists, state Tracked)."""
    b = bot(frame_rate=3, max_tracks=64, max_dets=8, feat_dim=16)             # max_time_lost = 30 frames
    ora = b.device_tracker.ora
    rng = np.random.default_rng(0)
    proto = rng.standard_normal((3, 16)).astype(np.float32)
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    boxes = np.asarray([[100, 100, 150, 220], [400, 100, 460, 230], [700, 300, 760, 420]], dtype=np.int32)
    zombie = None
    for f in range(40):
        vis = [0, 1, 2] if (f <= 2 or f >= 34) else [1, 2]                    # identity 0 hides for 31 frames, then returns
        out = b.update_arrays(boxes[vis], np.full(len(vis), 0.95, np.float32), proto[vis].copy())
        assert [t.track_id for t in out] == [t.track_id for t in ora.tracked]
        assert [t.track_id for t in b.lost_stracks] == [t.track_id for t in ora.lost]
        assert [t.track_id for t in b.removed_stracks] == [t.track_id for t in ora.removed], f
        assert [t.state for t in b.removed_stracks] == [t.state for t in ora.removed], f
        if f == 33:
            assert len(ora.removed) == 1 and ora.removed[0] in ora.lost, "the stream must expire identity 0 at frame 33"
            zombie = b.removed_stracks[0]
            assert zombie.state == pkg.TrackState.Removed and any(t is zombie for t in b.lost_stracks)
        if f == 34:
            assert ora.removed[0] in ora.tracked, "... and re-find it at frame 34"
            assert zombie.state == pkg.TrackState.Tracked and any(t is zombie for t in out) and b.removed_stracks[0] is zombie


def test_expired_lost_tracks_are_appended_twice_like_the_reference(bot):
    """D:1623-1627 runs over lost_stracks as the frame found it: a track that expired the frame before is still there (Q6), is
    marked removed and appended AGAIN, and only then filtered.  The recorded lifecycle stream has such repeats (and frames where
    a repeat and a first expiry interleave in lost-list order): removed_stracks must equal the reference's list on every frame."""
    import collections
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_lifecycle.npz"))
    K, M, D, frames = int(g["K"]), int(g["M"]), int(g["D"]), int(g["frames"])
    b = bot(frame_rate=int(g["frame_rate"]), max_tracks=4096, max_dets=max(K, M) + 8, feat_dim=D)   # (the double never recycles slots)
    st = pkg.synth.SyntheticStream(K, M, D, seed=int(g["seed"]), churn=float(g["churn"]))
    ora = b.device_tracker.ora
    for f in range(frames):
        fr = st.next()
        b.update_arrays(fr.boxes, fr.scores, fr.feats)
        assert len(ora.removed) == int(g["f%d_n_removed" % f]), "the oracle itself must follow the recorded reference"
        assert [t.track_id for t in b.removed_stracks] == [t.track_id for t in ora.removed], f
    twice = [k for k, v in collections.Counter(t.track_id for t in ora.removed).items() if v > 1]
    assert len(twice) >= 3, "the stream must exercise the repeats"
    for tid in twice:
        objs = [t for t in b.removed_stracks if t.track_id == tid]
        assert all(o is objs[0] for o in objs), "the reference appends the same object"


def test_drop_in_update_returns_plain_lists_update_arrays_lazy_ones(bot):
    """`BoTSORT.update(image)` -- the reference's own signature -- hands out the reference's type, a plain `list` a
    caller may append to / sort / `+=` / serialise; `update_arrays` (this package's entry point) keeps the lazy TrackList.
    `lazy_lists=True / False` force one form for both."""
    st = _stream()

    class Det:
        def __init__(self):
            self.fr = None

        def __call__(self, image):
            return [pkg.Box(classid=0, score=float(s), x1=int(b[0]), y1=int(b[1]), x2=int(b[2]), y2=int(b[3]), cx=0, cy=0) for b, s in zip(self.fr.boxes, self.fr.scores)]

    class Enc:
        feature_size = 32

        def __init__(self):
            self.fr = None

        def __call__(self, base_images, target_features):
            return None, self.fr.feats.copy()

    for lazy, want_update, want_arrays in ((None, list, trk.TrackList), (True, trk.TrackList, trk.TrackList), (False, list, list)):
        b = bot(max_tracks=512, max_dets=64, feat_dim=32, lazy_lists=lazy)
        b.detector, b.body_encoder = Det(), Enc()
        image = np.zeros((2000, 2000, 3), dtype=np.uint8)
        for f in range(3):
            fr = st.next()
            b.detector.fr = b.body_encoder.fr = fr
            out = b.update(image)
            assert type(out) is want_update and type(b.tracked_stracks) is want_update and type(b.lost_stracks) is want_update, (lazy, f)
            if want_update is list:
                out.append(None); out.pop()                      # a list is a list
                assert out == sorted(out, key=lambda t: t.track_id) or True
            fr = st.next()
            out = b.update_arrays(fr.boxes, fr.scores, fr.feats)
            assert type(out) is want_arrays and type(b.lost_stracks) is want_arrays, (lazy, f)
