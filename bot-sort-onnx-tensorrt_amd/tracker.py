"""`STrack` / `BoTSORT` with the reference's call surface (demo_bottrack_onnx_tflite.py:382-688,
1252-1639) over the resident HIP tracker (csrc/tracker.hip, C ABI `bts_tracker_*`).

The per-frame association cascade, Kalman filtering, feature bookkeeping and list merging all
run on the MI355X with no host round trip; this module only
  * hands the frame's detections to the device (`BoTSORT.update_arrays`, or `update(image)` which
    first calls the -- unchanged, out-of-scope -- detector / ReID encoders like the reference),
  * and, for drop-in compatibility, mirrors the resulting track lists: ONE export call per frame lands both lists in
    pinned host memory (`bts_tracker_export_lists`); `tracked_stracks` / `lost_stracks` are `TrackList`s over that
    snapshot which build an `STrack` mirror only for the entries the caller touches (D:2135-2185 reads `tlbr`,
    `track_id`, `body`); a mirror the caller keeps across frames re-reads its row on first access after each frame.
`update_arrays(..., export=False)` skips that mirror for throughput-critical loops.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict, deque
from collections.abc import Sequence
from typing import List, Optional

import numpy as np

from . import _lib, matching
from .kalman_filter import KalmanFilter


class Box:                             # D:84-95: what the detector hands over, one per detection
    def __init__(self, trackid=0, classid=0, score=0.0, x1=0, y1=0, x2=0, y2=0, cx=0, cy=0, is_used=False):
        self.trackid, self.classid, self.score = trackid, classid, score
        self.x1, self.y1, self.x2, self.y2, self.cx, self.cy = x1, y1, x2, y2, cx, cy
        self.is_used = is_used

    @classmethod
    def of(cls, box, **extra):
        return cls(trackid=box.trackid, classid=box.classid, score=box.score, x1=box.x1, y1=box.y1, x2=box.x2, y2=box.y2,
                   cx=box.cx, cy=box.cy, is_used=False, **extra)


class Body(Box):                       # D:97-102 (parameter order as in the reference: the ten Box fields first)
    def __init__(self, trackid=0, classid=0, score=0.0, x1=0, y1=0, x2=0, y2=0, cx=0, cy=0, is_used=False,
                 head=None, hand1=None, hand2=None):
        super().__init__(trackid, classid, score, x1, y1, x2, y2, cx, cy, is_used)
        self.head, self.hand1, self.hand2 = head, hand1, hand2


class Head(Box):                       # D:104-108
    def __init__(self, trackid=0, classid=0, score=0.0, x1=0, y1=0, x2=0, y2=0, cx=0, cy=0, is_used=False,
                 face=None, face_landmarks=None):
        super().__init__(trackid, classid, score, x1, y1, x2, y2, cx, cy, is_used)
        self.face, self.face_landmarks = face, face_landmarks


class Face(Box):                       # D:110-112
    pass


class Hand(Box):                       # D:114-116
    pass


def attach_parts(detected_boxes):
    """D:1303-1411: split one frame's detections by class (0 body, 1 head, 2 hand, 3 face) and attach
    face -> head, head -> body, two hands -> body with the greedy rule of find_most_relevant_object,
    one device call per pass.  Returns (bodies, heads, hands, faces)."""
    bodies = [Body.of(b) for b in detected_boxes if b.classid == 0]
    heads = [Head.of(b) for b in detected_boxes if b.classid == 1]
    hands = [Hand.of(b) for b in detected_boxes if b.classid == 2]
    faces = [Face.of(b) for b in detected_boxes if b.classid == 3]

    def attach(bases, targets, picks, setter):
        if not bases or not targets:
            return
        xyxy = lambda L: [[o.x1, o.y1, o.x2, o.y2] for o in L]
        cen = lambda L: [[o.cx, o.cy] for o in L]
        idx, used = matching.associate_parts(xyxy(bases), cen(bases), xyxy(targets), cen(targets), picks=picks)
        for t, u in zip(targets, used):
            t.is_used = bool(u)
        for b, row in zip(bases, idx):
            for p, k in enumerate(row):
                if k >= 0:
                    setter(b, p, targets[int(k)])

    attach(heads, faces, 1, lambda h, p, f: setattr(h, "face", f))
    attach(bodies, heads, 1, lambda b, p, h: setattr(b, "head", h))
    attach(bodies, hands, 2, lambda b, p, h: setattr(b, "hand1" if p == 0 else "hand2", h))
    return bodies, heads, hands, faces


class TrackState:                      # D:382-387
    New = 0
    Tracked = 1
    Lost = 2
    LongLost = 3
    Removed = 4


class BaseTrack:                       # D:389-437 (the ID counter lives in the device tracker)
    _count = 0
    track_id = 0
    is_activated = False
    state = TrackState.New
    history = OrderedDict()
    features = []
    score = 0
    start_frame = 0
    frame_id = 0
    time_since_update = 0
    location = (np.inf, np.inf)           # (multi-camera field of the reference; never read by the tracker)

    @property
    def end_frame(self):
        return self.frame_id

    @staticmethod
    def next_id():
        BaseTrack._count += 1
        return BaseTrack._count

    @staticmethod
    def clear_count():
        BaseTrack._count = 0

    def activate(self, *args):             # abstract in the reference too (D:389-437): STrack overrides all three
        raise NotImplementedError

    def predict(self):
        raise NotImplementedError

    def update(self, *args, **kwargs):
        raise NotImplementedError

    def mark_lost(self):
        self.state = TrackState.Lost

    def mark_long_lost(self):
        self.state = TrackState.LongLost

    def mark_removed(self):
        self.state = TrackState.Removed


def _lazy(field, name):
    """A per-track array that lives in the device table of the owning BoTSORT (covariance, ReID feature rows): a mirror
    fetches it on first access after a frame (bts_tracker_read_slot) and caches it until the next frame; standalone
    STracks (detection records, tracks driven by hand through activate / update) simply hold their own value.  A tracker
    without face features never asks the device for them: the attribute stays None like the reference's (D:481-485)."""
    attr = "_" + name

    def get(self):
        d = self.__dict__
        own = d.get("_owner")
        if own is not None and d["_live"] and not (field >= 4 and own.face_dim == 0):
            if d["_epoch"] != own._epoch:
                own._refresh(self)
            if d["_live"] and d["slot"] >= 0 and d["_fresh"].get(name) != own._epoch:
                d[attr] = own.device_tracker.read_slot(d["slot"], field)
                d["_fresh"][name] = own._epoch
        return d.get(attr)

    def set(self, value):
        d = self.__dict__
        d[attr] = value
        if d.get("_owner") is not None:
            d["_fresh"][name] = d["_owner"]._epoch

    return property(get, set)


def _mirrored(name, default):
    """A field the device refreshes every frame (state, frame counters, mean, ...).  Standalone STracks hold a plain value;
    a mirror owned by a BoTSORT re-reads its row of the current snapshot on the first access after a frame, so an object the
    caller kept from an earlier frame shows the track's current state like the reference's mutated-in-place STrack does."""
    def get(self):
        d = self.__dict__
        own = d.get("_owner")
        if own is not None and d["_epoch"] != own._epoch and d["_live"]:
            own._refresh(self)
        return d.get(name, default)

    def set(self, value):
        self.__dict__[name] = value

    return property(get, set)


class STrack(BaseTrack):
    """One track or one detection record, D:439-688.

    Two lives.  (1) Mirror of a track owned by a `BoTSORT`: Kalman state and ReID features live in the device track
    table; the integer fields, `mean` and `tlbr` are refreshed after every frame, `covariance` and the feature rows are
    read from the table on access.  (2) Standalone object driven by hand exactly like the reference's
    (`STrack(tlwh, score, ...)`, `activate`, `predict`, `update`, `re_activate`, `update_*_features`): the arithmetic
    runs through the standalone C-ABI kernels (`KalmanFilter`, `bts_feature_ema`) on the arrays the object holds."""
    shared_kalman = None

    mean = _mirrored("mean", None)
    state = _mirrored("state", TrackState.New)
    is_activated = _mirrored("is_activated", False)
    frame_id = _mirrored("frame_id", 0)
    start_frame = _mirrored("start_frame", 0)
    tracklet_len = _mirrored("tracklet_len", 0)
    score = _mirrored("score", 0)
    det_index = _mirrored("det_index", -1)
    slot = _mirrored("slot", -1)
    _dev_tlbr = _mirrored("_dev_tlbr", None)

    def __init__(self, tlwh, score, feature_history=300, body=None, body_feature=None, face_feature=None):
        self._owner, self._live, self._fresh, self._epoch = None, True, {}, -1
        self._tlwh = np.asarray(tlwh, dtype=np.float32)            # D:465
        self.kalman_filter = None
        self.mean, self._covariance = None, None
        self.is_activated = False
        self.score = score
        self.tracklet_len = 0
        self.alpha = 0.9                                           # D:473
        self.body = body
        self._body_curr_feature = None                             # D:475-490
        self._body_smooth_feature = None
        self._face_curr_feature = None
        self._face_smooth_feature = None
        self.feature_history = feature_history
        self.body_features = deque([], maxlen=feature_history)
        self.face_features = deque([], maxlen=feature_history)
        self.det_index = -1
        self.slot = -1
        self._dev_tlbr = None       # STrack.tlbr as evaluated on the device (float32 flow for fresh tracks)
        if body_feature is not None:
            self.update_body_features(body_feature)                # D:477-478
        if face_feature is not None:
            self.update_face_features(face_feature)                # D:484-485

    # ---- mirror of a device-resident track (built by BoTSORT, never by the caller) ---------------------------------
    _ZERO_TLWH = np.zeros(4, dtype=np.float32)

    @classmethod
    def _new_mirror(cls, owner, track_id):
        """A mirror without running __init__ (no deques, no arrays: thousands may be built per frame)."""
        st = cls.__new__(cls)
        st.__dict__.update(_owner=owner, _live=True, _fresh={}, _epoch=-1, _tlwh=cls._ZERO_TLWH, kalman_filter=owner.kalman_filter,
                           _covariance=None, alpha=0.9, body=None, _body_curr_feature=None, _body_smooth_feature=None,
                           _face_curr_feature=None, _face_smooth_feature=None, feature_history=owner.feature_history,
                           track_id=track_id, slot=-1)
        return st

    def __getattr__(self, name):                   # (only reached when normal lookup fails: the mirrors' feature deques)
        if name in ("body_features", "face_features") and "_owner" in self.__dict__:
            dq = deque([], maxlen=self.__dict__.get("feature_history", 300))
            self.__dict__[name] = dq
            return dq
        raise AttributeError(name)

    def _bind(self, snap, which, k):
        """Row k of list `which` of a snapshot becomes this mirror's state (the per-frame fields of D:439-610)."""
        d = self.__dict__
        i = snap.ints[which][k].tolist()
        d["state"], d["is_activated"], d["frame_id"], d["start_frame"] = i[1], bool(i[2]), i[3], i[4]
        d["tracklet_len"], d["det_index"], d["slot"] = i[5], i[6], i[7]
        d["mean"] = snap.mean[which][k]
        d["_dev_tlbr"] = snap.tlbr[which][k]
        d["score"] = float(snap.score[which][k])
        d["_epoch"] = snap.epoch

    covariance = _lazy(1, "covariance")
    body_curr_feature = _lazy(2, "body_curr_feature")
    body_smooth_feature = _lazy(3, "body_smooth_feature")
    face_curr_feature = _lazy(4, "face_curr_feature")
    face_smooth_feature = _lazy(5, "face_smooth_feature")

    # ---- ReID feature bookkeeping, D:492-514 (quirk Q8: the first call aliases smooth = feature and normalises IN PLACE)
    def _update_features(self, which, feature):
        feature = np.asarray(feature)
        if not (feature.dtype == np.float32 and feature.flags.c_contiguous and feature.flags.writeable):
            feature = np.array(feature, dtype=np.float32)          # (a private copy: nothing of the caller's to alias)
        smooth = getattr(self, which + "_smooth_feature")
        new_smooth, feature = matching.feature_ema(smooth, feature, self.alpha)
        setattr(self, which + "_curr_feature", feature)
        setattr(self, which + "_smooth_feature", new_smooth)
        getattr(self, which + "_features").append(feature)

    def update_body_features(self, feature):
        self._update_features("body", feature)

    def update_face_features(self, feature):
        self._update_features("face", feature)

    # ---- Kalman life cycle, D:516-610
    def predict(self):
        """D:516-522 (dead code in the reference's loop, kept for API parity)."""
        mean_state = self.mean.copy()
        if self.state != TrackState.Tracked:
            mean_state[6] = 0
            mean_state[7] = 0
        self.mean, self.covariance = self.kalman_filter.predict(mean_state, self.covariance)

    def activate(self, kalman_filter, frame_id):
        """D:556-568: start a new tracklet."""
        self.kalman_filter = kalman_filter
        self.track_id = self.next_id()
        self.mean, self.covariance = self.kalman_filter.initiate(self.tlwh_to_xywh(self._tlwh))
        self._dev_tlbr = None
        self.tracklet_len = 0
        self.state = TrackState.Tracked
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = frame_id
        self.start_frame = frame_id

    def re_activate(self, new_track, frame_id, new_id=False):
        """D:570-584."""
        self.mean, self.covariance = self.kalman_filter.update(self.mean, self.covariance, self.tlwh_to_xywh(new_track.tlwh))
        self._dev_tlbr = None
        if new_track.body_curr_feature is not None:
            self.update_body_features(new_track.body_curr_feature)
        if new_track.face_curr_feature is not None:
            self.update_face_features(new_track.face_curr_feature)
        self.tracklet_len = 0
        self.state = TrackState.Tracked
        self.is_activated = True
        self.frame_id = frame_id
        if new_id:
            self.track_id = self.next_id()
        self.score = new_track.score
        self.body = new_track.body

    def update(self, new_track, frame_id):
        """D:586-610: update a matched track."""
        self.frame_id = frame_id
        self.tracklet_len += 1
        new_tlwh = new_track.tlwh
        self.mean, self.covariance = self.kalman_filter.update(self.mean, self.covariance, self.tlwh_to_xywh(new_tlwh))
        self._dev_tlbr = None
        if new_track.body_curr_feature is not None:
            self.update_body_features(new_track.body_curr_feature)
        if new_track.face_curr_feature is not None:
            self.update_face_features(new_track.face_curr_feature)
        self.state = TrackState.Tracked
        self.is_activated = True
        self.score = new_track.score
        self.body = new_track.body

    # ---- box formats, D:624-685
    @property
    def tlwh(self):
        if self.mean is None:
            return self._tlwh.copy()
        ret = self.mean[:4].copy()
        ret[:2] -= ret[2:] / 2
        return ret

    @property
    def tlbr(self):
        if self._dev_tlbr is not None:
            return self._dev_tlbr.copy()
        ret = self.tlwh.copy()
        ret[2:] += ret[:2]
        return ret

    @property
    def xywh(self):
        ret = self.tlwh.copy()
        ret[:2] += ret[2:] / 2.0
        return ret

    @staticmethod
    def tlwh_to_xyah(tlwh):
        ret = np.asarray(tlwh).copy()
        ret[:2] += ret[2:] / 2
        ret[2] /= ret[3]
        return ret

    @staticmethod
    def tlwh_to_xywh(tlwh):
        ret = np.asarray(tlwh).copy()
        ret[:2] += ret[2:] / 2
        return ret

    def to_xywh(self):
        return self.tlwh_to_xywh(self.tlwh)

    @staticmethod
    def tlbr_to_tlwh(tlbr):
        ret = np.asarray(tlbr).copy()
        ret[2:] -= ret[:2]
        return ret

    @staticmethod
    def tlwh_to_tlbr(tlwh):
        ret = np.asarray(tlwh).copy()
        ret[2:] += ret[:2]
        return ret

    @staticmethod
    def multi_gmc(stracks: "List[STrack]", H=np.eye(2, 3)):
        """D:538-554 on host mirrors that carry mean AND covariance (API parity).  Tracks owned by a
        BoTSORT live on the device: use `BoTSORT.apply_gmc(H)` for those."""
        if len(stracks) == 0:
            return
        kf = STrack.shared_kalman or KalmanFilter()
        STrack.shared_kalman = kf
        mean, cov = kf.multi_gmc(np.asarray([st.mean for st in stracks]), np.asarray([st.covariance for st in stracks]), H)
        for i, st in enumerate(stracks):
            st.mean, st.covariance = mean[i], cov[i]

    @staticmethod
    def multi_predict(stracks: "List[STrack]"):
        """D:524-536 on host mirrors (API parity; the tracker predicts on the device table)."""
        if len(stracks) == 0:
            return
        kf = STrack.shared_kalman or KalmanFilter()
        STrack.shared_kalman = kf
        mean = np.asarray([st.mean.copy() for st in stracks])
        cov = np.asarray([st.covariance for st in stracks])
        zero = np.asarray([st.state != TrackState.Tracked for st in stracks])
        mean, cov = kf.multi_predict(mean, cov, zero_velocity=zero)
        for i, st in enumerate(stracks):
            st.mean, st.covariance = mean[i], cov[i]

    def propagate_trackid_to_related_objects(self):                      # D:612-622
        body = self.body
        if body is None:
            return
        parts = [body, getattr(body, "head", None), getattr(body, "hand1", None), getattr(body, "hand2", None)]
        if parts[1] is not None:
            parts.append(getattr(parts[1], "face", None))
        for o in parts:
            if o is not None and hasattr(o, "trackid"):
                o.trackid = self.track_id

    def __repr__(self):
        return "OT_{}_({}-{})".format(self.track_id, self.start_frame, self.end_frame)   # D:687-688


def forms_string(forms) -> str:
    """"KEY=VALUE,..." for bts_tracker_create_forms from a dict or a string.  A dict value of False / None / "" becomes "KEY=" --
    the form switched OFF whatever the environment says.  An int 0 stays "KEY=0": the library reads "0" as off for the boolean
    forms, and for a numeric key ({"BTS_LAP_HELPERS": 0} = no helper workgroups) it is a value, not "use the default"."""
    if isinstance(forms, dict):
        return ",".join("%s=%s" % (k, "" if (v is None or v is False or v == "") else (1 if v is True else v)) for k, v in forms.items())
    return str(forms) if forms else ""


def tracker_memory_bytes(max_tracks: int, max_dets: int, feat_dim: int, face_dim: int = 0, record_costs: bool = False, forms=None, **options) -> int:
    """Device bytes a DeviceTracker with these capacities (and execution forms) allocates (bts_tracker_memory_bytes[_forms]:
    host-only, no GPU needed)."""
    lib = _lib.load()
    cfg = _lib.TrackerConfig()
    _lib.check(lib.bts_tracker_default_config(ctypes.byref(cfg)))
    cfg.max_tracks, cfg.max_dets, cfg.feat_dim, cfg.face_dim = max_tracks, max_dets, feat_dim, face_dim
    cfg.record_costs = 1 if record_costs else 0
    for k, v in options.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown tracker option %r" % k)
        setattr(cfg, k, v)
    out = ctypes.c_uint64()
    forms = forms_string(forms)
    if forms:
        _lib.check(lib.bts_tracker_memory_bytes_forms(ctypes.byref(cfg), str(forms).encode(), ctypes.byref(out)))
    else:
        _lib.check(lib.bts_tracker_memory_bytes(ctypes.byref(cfg), ctypes.byref(out)))
    return int(out.value)


COUNT_KEYS = ("frame_id", "n_tracked", "n_lost", "n_removed", "last_id", "n_pool", "n_hi", "n_lo", "n_unconfirmed",
              "n_stage2_rows", "n_stage3_cols", "errors")


class Snapshot:
    """One frame's tracked (0) and lost (1) lists as column arrays, list order: ints[which] [n,8] (track_id, state,
    is_activated, frame_id, start_frame, tracklet_len, det_index, slot), mean [n,8], tlbr [n,4], score [n]."""
    __slots__ = ("epoch", "counts", "ints", "mean", "tlbr", "score", "_ids", "_where")

    def __init__(self, epoch, counts12, ints, mean, tlbr, score):
        self.epoch, self.ints, self.mean, self.tlbr, self.score = epoch, ints, mean, tlbr, score
        self.counts = dict(zip(COUNT_KEYS, (int(v) for v in counts12)))
        self._ids, self._where = [None, None], None

    def n(self, which):
        return len(self.ints[which])

    def ids(self, which):
        """track ids of a list as Python ints (converted once per frame, on first use)."""
        if self._ids[which] is None:
            self._ids[which] = self.ints[which][:, 0].tolist()
        return self._ids[which]

    def locate(self, track_id):
        """(which, row) of a track id, or None: the index is built on the first lookup of a frame (only mirrors the caller
        kept from earlier frames need it -- list access binds by position)."""
        if self._where is None:
            w = {}
            for which in (1, 0):
                w.update(zip(self.ids(which), ((which, k) for k in range(self.n(which)))))
            self._where = w
        return self._where.get(track_id)


class TrackList(Sequence):
    """`tracked_stracks` / `lost_stracks` of one frame (D:1629-1639) over a Snapshot: len(), indexing, slicing and iteration
    like the reference's lists, but the STrack mirror of an entry is only built when the entry is touched."""
    __slots__ = ("_owner", "_snap", "_which", "_n", "_items")

    def __init__(self, owner, snap, which):
        self._owner, self._snap, self._which = owner, snap, which
        self._n = snap.n(which)
        self._items = [None] * self._n

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self._n))]
        if k < 0:
            k += self._n
        if not 0 <= k < self._n:
            raise IndexError("list index out of range")
        st = self._items[k]
        if st is None:
            st = self._items[k] = self._owner._mirror_at(self._snap, self._which, k)
        return st

    def __iter__(self):
        for k in range(self._n):
            yield self[k]

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __eq__(self, other):
        return isinstance(other, (list, TrackList)) and len(other) == self._n and all(a is b for a, b in zip(self, other))

    def copy(self):
        return list(self)

    def __repr__(self):
        return "TrackList(%d %s)" % (self._n, "tracked" if self._which == 0 else "lost")


class DeviceTracker:
    """Thin owner of a `bts_tracker` handle."""

    def __init__(self, max_tracks: int, max_dets: int, feat_dim: int, face_dim: int = 0, frame_rate: int = 30,
                 record_costs: bool = False, device: Optional[int] = None, context: Optional[_lib.Context] = None,
                 forms=None, **thresholds):
        """`forms`: the execution forms of this tracker (bts_tracker_create_forms) -- "KEY=VALUE,..." or a dict; None = the process
        environment / the defaults.  Forms never change a result; `tune.select_forms` picks the fastest set on the caller's frames."""
        lib = _lib.load(require_gpu=True)
        self.ctx = context or _lib.default_context(device)
        cfg = _lib.TrackerConfig()
        _lib.check(lib.bts_tracker_default_config(ctypes.byref(cfg)))
        cfg.max_tracks, cfg.max_dets, cfg.feat_dim, cfg.face_dim = max_tracks, max_dets, feat_dim, face_dim
        cfg.frame_rate = frame_rate
        cfg.record_costs = 1 if record_costs else 0
        for k, v in thresholds.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown tracker option %r" % k)
            setattr(cfg, k, v)
        self.cfg = cfg
        h = ctypes.c_void_p()
        forms = forms_string(forms)
        if forms:
            _lib.check(lib.bts_tracker_create_forms(self.ctx.handle, ctypes.byref(cfg), str(forms).encode(), ctypes.byref(h)))
        else:
            _lib.check(lib.bts_tracker_create(self.ctx.handle, ctypes.byref(cfg), ctypes.byref(h)))
        self.forms_requested = str(forms) if forms else ""
        self.handle = h
        self._lib = lib
        t = max_tracks
        self._ints = np.empty((t, 8), dtype=np.int32)
        self._mean = np.empty((t, 8), dtype=np.float64)
        self._tlbr = np.empty((t, 4), dtype=np.float64)
        self._score = np.empty((t,), dtype=np.float32)
        self._pin = None

    def reset(self):
        _lib.check(self._lib.bts_tracker_reset(self.handle, _lib.current_stream()))

    def forms(self) -> dict:
        """The execution forms in effect (bts_tracker_forms), e.g. {'edges': 'tiles', 'lists_ahead': '0', 'lap_form': 'col', ...}."""
        buf = ctypes.create_string_buffer(512)
        _lib.check(self._lib.bts_tracker_forms(self.handle, buf, 512))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(",") if "=" in kv)

    def step(self, boxes, scores, feats, face_feats=None):
        """boxes int32 [M,4] (tlbr pixels), scores float32 [M], feats float32 [M,D] -- CUDA tensors.
        Enqueues one BoTSORT.update on the current stream; returns immediately."""
        m = int(boxes.shape[0])
        _lib.check(self._lib.bts_tracker_step(self.handle, _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(feats),
                                              _lib.ptr(face_feats), m, _lib.current_stream()))

    def inputs_resident(self, flag: bool = True):
        """Promise (bts_tracker_inputs_resident) that the tensors given to the following step() calls are complete in device
        memory when step() is called -- nothing pending on any stream still writes them.  bench.py makes it for its frames
        (uploaded and synchronised before the timed region); the BoTSORT facade, which uploads per call, does not."""
        _lib.check(self._lib.bts_tracker_inputs_resident(self.handle, 1 if flag else 0))

    def export_lists(self, epoch: int = 0) -> "Snapshot":
        """Both lists in one call (bts_tracker_export_lists): the export kernels write into pinned host buffers owned by
        this object, one stream synchronisation; the rows are copied out, so a Snapshot stays valid when the next frame
        is exported.  Raises BtsError when the device flagged a truncated frame."""
        if self._pin is None:
            t = self.cfg.max_tracks
            spec = ((np.int32, 8), (np.float64, 8), (np.float64, 4), (np.float32, 1))
            ptrs, views = [], []
            for dt, w in spec:
                pp = ctypes.c_void_p()
                _lib.check(self._lib.bts_pinned_alloc(ctypes.byref(pp), 2 * t * w * np.dtype(dt).itemsize))
                ptrs.append(pp)
                ct = {np.int32: ctypes.c_int32, np.float64: ctypes.c_double, np.float32: ctypes.c_float}[dt]
                a = np.ctypeslib.as_array(ctypes.cast(pp, ctypes.POINTER(ct)), shape=(2 * t, w))
                views.append(a if w > 1 else a.reshape(2 * t))
            self._pin = (ptrs, views)
        ptrs, (vi, vm, vb, vs) = self._pin
        c = np.zeros(12, dtype=np.int32)
        _lib.check(self._lib.bts_tracker_export_lists(self.handle, ptrs[0], ptrs[1], ptrs[2], ptrs[3], c.ctypes.data, _lib.current_stream()))
        t, n0, n1 = self.cfg.max_tracks, int(c[1]), int(c[2])
        return Snapshot(epoch, c, [vi[:n0].copy(), vi[t:t + n1].copy()], [vm[:n0].copy(), vm[t:t + n1].copy()],
                        [vb[:n0].copy(), vb[t:t + n1].copy()], [vs[:n0].copy(), vs[t:t + n1].copy()])

    def read_column(self, field: int) -> np.ndarray:
        """int32 [max_tracks], slot order: field 0 track_id, 1 state, 2 frame_id (bts_tracker_read_column)."""
        out = np.empty(self.cfg.max_tracks, dtype=np.int32)
        _lib.check(self._lib.bts_tracker_read_column(self.handle, int(field), out.ctypes.data, _lib.current_stream()))
        return out

    def export_mot(self) -> np.ndarray:
        """MOTChallenge rows [n,10] of the tracked list: frame, id, left, top, width, height, conf, -1, -1, -1."""
        buf = np.empty((self.cfg.max_tracks, 10), dtype=np.float64)
        n = _lib.check(self._lib.bts_tracker_export_mot(self.handle, buf.ctypes.data, _lib.current_stream()))
        return buf[:n].copy()

    def export_mot_async(self, buf: "PinnedMot"):
        """Enqueue the MOT rows of the tracked list into `buf` (pinned host memory) behind the steps already on the
        current stream -- no host synchronisation; read `buf.rows()` after an event / sync of your own."""
        _lib.check(self._lib.bts_tracker_export_mot_async(self.handle, buf.rows_ptr, buf.n_ptr, _lib.current_stream()))

    def export_state(self, which: int, cov=True, body=True, face=False) -> dict:
        """covariance [n,8,8] / body_curr, body_smooth [n,D] / face_* [n,Df] of the tracked (0) or lost (1) list, list order."""
        t, D, Df = self.cfg.max_tracks, self.cfg.feat_dim, self.cfg.face_dim
        n = self.counts()["n_tracked" if which == 0 else "n_lost"]
        bufs = [np.empty((n, 64), np.float64) if cov else None,
                np.empty((n, D), np.float32) if body else None, np.empty((n, D), np.float32) if body else None,
                np.empty((n, Df), np.float32) if face and Df else None, np.empty((n, Df), np.float32) if face and Df else None]
        got = _lib.check(self._lib.bts_tracker_export_state(self.handle, which, *[b.ctypes.data if b is not None and b.size else None for b in bufs],
                                                            _lib.current_stream()))
        assert got == n
        keys = ("covariance", "body_curr_feature", "body_smooth_feature", "face_curr_feature", "face_smooth_feature")
        out = {k: b for k, b in zip(keys, bufs) if b is not None}
        if "covariance" in out:
            out["covariance"] = out["covariance"].reshape(n, 8, 8)
        return out

    def read_slot(self, slot: int, field: int) -> np.ndarray:
        """One table row: field 0 mean, 1 covariance [8,8], 2/3 body curr/smooth, 4/5 face curr/smooth."""
        D, Df = self.cfg.feat_dim, self.cfg.face_dim
        if field == 6:
            out = np.empty(2, np.int32)
            _lib.check(self._lib.bts_tracker_read_slot(self.handle, int(slot), 6, out.ctypes.data, _lib.current_stream()))
            return out
        out = np.empty(8 if field == 0 else 64 if field == 1 else D if field <= 3 else Df, np.float64 if field <= 1 else np.float32)
        _lib.check(self._lib.bts_tracker_read_slot(self.handle, int(slot), int(field), out.ctypes.data, _lib.current_stream()))
        return out.reshape(8, 8) if field == 1 else out

    def debug_filter(self) -> dict:
        """Counters of the appearance pre-filter for the last step (bts_tracker_debug_filter)."""
        h = np.zeros(16, dtype=np.int32)
        _lib.check(self._lib.bts_tracker_debug_filter(self.handle, h.ctypes.data, _lib.current_stream()))
        self._filter_raw = h
        return {"k_stages": int(h[0]), "tiles": int(h[1]), "prefix_exits": int(h[2]), "candidates": int(h[3]), "tests": int(h[4]),
                "tile": (int(h[14]) or 128, int(h[15]) or 128),      # rows x columns of a tile: a K stage is that x 128 elements (BTS_FILTER_TILE)
                # clock cycles summed over the persistent workgroups (one wave's view): candidate tests, waits for operand stages, total
                "cycles_tests": int(h[5]) * 64, "cycles_stage_waits": int(h[6]) * 64, "cycles_total": int(h[7]) * 64}

    def debug_lap(self, stage: int) -> np.ndarray:
        """Counters of the assignment solver for stage 1..3 of the last step (bts_tracker_debug_lap)."""
        h = np.zeros(64, dtype=np.int32)
        _lib.check(self._lib.bts_tracker_debug_lap(self.handle, int(stage), h.ctypes.data, _lib.current_stream()))
        return h

    def gmc(self, H):
        """STrack.multi_gmc (D:538-554) on every live track; H is a 2x3 affine warp."""
        h = np.ascontiguousarray(H, dtype=np.float64).reshape(6)
        _lib.check(self._lib.bts_tracker_gmc(self.handle, h.ctypes.data, _lib.current_stream()))

    def counts(self) -> dict:
        c = np.zeros(12, dtype=np.int32)
        _lib.check(self._lib.bts_tracker_counts(self.handle, c.ctypes.data, _lib.current_stream()))
        return dict(zip(COUNT_KEYS, (int(v) for v in c)))

    def export(self, which: int) -> dict:
        n = _lib.check(self._lib.bts_tracker_export(self.handle, which, self._ints.ctypes.data, self._mean.ctypes.data,
                                                    self._tlbr.ctypes.data, self._score.ctypes.data,
                                                    _lib.current_stream()))
        i = self._ints[:n]
        return {"track_id": i[:, 0].astype(np.int64), "state": i[:, 1].astype(np.int64),
                "is_activated": i[:, 2].astype(np.int64), "frame_id": i[:, 3].astype(np.int64),
                "start_frame": i[:, 4].astype(np.int64), "tracklet_len": i[:, 5].astype(np.int64),
                "det_index": i[:, 6].astype(np.int64), "slot": i[:, 7].astype(np.int64),
                "mean": self._mean[:n].copy(), "tlbr": self._tlbr[:n].copy(), "score": self._score[:n].copy()}

    def stage_result(self, stage: int, want_cost: bool = False) -> dict:
        nr, nc = ctypes.c_int32(0), ctypes.c_int32(0)
        t, m = self.cfg.max_tracks, self.cfg.max_dets
        x = np.empty(t, dtype=np.int32)
        y = np.empty(m, dtype=np.int32)
        cost = None
        if want_cost:
            # sizes first, then the matrix
            _lib.check(self._lib.bts_tracker_stage_result(self.handle, stage, None, None, None, ctypes.byref(nr),
                                                          ctypes.byref(nc), _lib.current_stream()))
            cost = np.empty((nr.value, nc.value), dtype=np.float64)
        _lib.check(self._lib.bts_tracker_stage_result(self.handle, stage, x.ctypes.data, y.ctypes.data,
                                                      cost.ctypes.data if cost is not None and cost.size else None,
                                                      ctypes.byref(nr), ctypes.byref(nc), _lib.current_stream()))
        x, y = x[:nr.value], y[:nc.value]
        rows = np.nonzero(x >= 0)[0]
        return {"x": x, "y": y, "m": np.stack([rows, x[rows]], axis=1).astype(np.int64).reshape(-1, 2),
                "u_trk": np.nonzero(x < 0)[0].astype(np.int64), "u_det": np.nonzero(y < 0)[0].astype(np.int64),
                "cost": cost}

    SEGMENTS = {"prep": 0, "predict": 1, "cos1": 2, "cost1": 3, "lap1": 4, "apply1": 5, "stage2": 6, "stage3": 7,
                "merge": 8, "dedupe": 9, "k_filter": 20, "k_lap": 21, "k_edges1": 23, "k_mid2": 24, "k_apply23": 25,
                "step": 100}

    def profile_enable(self, segment: str, capacity: int, stride: int = 1):
        """hipEvent pairs around `segment` on every `stride`-th step, at most `capacity` of them."""
        _lib.check(self._lib.bts_tracker_profile_enable(self.handle, self.SEGMENTS[segment], int(capacity)))
        _lib.check(self._lib.bts_tracker_profile_stride(self.handle, int(stride)))

    def profile_read(self):
        """-> (total_ms, count) of the hipEvent pairs recorded since the last read."""
        tot, cnt = ctypes.c_double(0.0), ctypes.c_int32(0)
        _lib.check(self._lib.bts_tracker_profile_read(self.handle, ctypes.byref(tot), ctypes.byref(cnt)))
        return tot.value, cnt.value

    def close(self):
        if getattr(self, "_pin", None):
            for pp in self._pin[0]:
                self._lib.bts_pinned_free(pp)
            self._pin = None
        if getattr(self, "handle", None):
            self._lib.bts_tracker_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedMot:
    """A pinned (device-mapped) host buffer for `DeviceTracker.export_mot_async`: max_tracks x 10 float64 rows + counters."""

    def __init__(self, max_tracks: int):
        lib = _lib.load(require_gpu=True)
        self._lib = lib
        self.max_tracks = int(max_tracks)
        pr, pn = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.bts_pinned_alloc(ctypes.byref(pr), self.max_tracks * 10 * 8))
        _lib.check(lib.bts_pinned_alloc(ctypes.byref(pn), 16))
        self.rows_ptr, self.n_ptr = pr, pn
        self._rows = np.ctypeslib.as_array(ctypes.cast(pr, ctypes.POINTER(ctypes.c_double)), shape=(self.max_tracks, 10))
        self._n = np.ctypeslib.as_array(ctypes.cast(pn, ctypes.POINTER(ctypes.c_int32)), shape=(4,))
        self._n[:] = 0

    def count(self) -> int:
        return int(self._n[0])

    def frame_id(self) -> int:
        return int(self._n[1])

    def errors(self) -> int:
        return int(self._n[2])

    def rows(self) -> np.ndarray:
        """The frame's rows (a view into the pinned buffer: copy before the buffer is reused)."""
        return self._rows[: self.count()]

    def close(self):
        if getattr(self, "rows_ptr", None):
            self._lib.bts_pinned_free(self.rows_ptr)
            self._lib.bts_pinned_free(self.n_ptr)
            self.rows_ptr = self.n_ptr = None
            self._rows = self._n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BoTSORT:
    """D:1252-1639.  `update(image)` keeps the reference signature; `update_arrays` is the
    tracker-only entry point fed with pre-extracted detections (BASELINE.json north_star)."""

    def __init__(self, object_detection_model=None, body_feature_extractor_model=None,
                 face_feature_extractor_model=None, frame_rate: int = 30, *, max_tracks: int = 4096,
                 max_dets: int = 1024, feat_dim: Optional[int] = None, face_dim: Optional[int] = None,
                 record_costs: bool = False, device: Optional[int] = None, lazy_lists: Optional[bool] = None,
                 keep_removed_state: bool = True, **tracker_options):
        """`tracker_options`: further `bts_tracker_config` fields by name (`motion_gate=1`, `appearance_mode=1`, ...).
        `lazy_lists`: None (default) = the drop-in call `update(image)` hands out plain Python `list`s like the reference (a
        caller may append / sort / `+=` / serialise them), while `update_arrays` -- this package's own entry point --
        hands out lazy TrackLists (a Sequence whose STrack mirrors are built on access: no per-track Python per frame);
        True / False force one form for both.  `keep_removed_state=False` skips reading a removed track's covariance /
        feature rows off the device."""
        self.lazy_lists, self.keep_removed_state = lazy_lists, bool(keep_removed_state)
        self._plain_call = False        # set by update(image) around its update_arrays call
        self._epoch, self._snap = 0, None
        self._removed_live = set()      # ids appended to removed_stracks that are still on a list (Q6: for one more frame)
        self._repeats = []              # (id, mirror) of expired lost tracks the reference appends a second time this frame
        self.tracked_stracks: List[STrack] = []
        self.lost_stracks: List[STrack] = []
        self.removed_stracks: List[STrack] = []
        BaseTrack.clear_count()                                   # D:1264
        self.frame_id = 0
        self.track_high_thresh = 0.40                             # D:1268-1277
        self.track_low_thresh = 0.1
        self.new_track_thresh = 0.9
        self.match_thresh = 0.8
        self.track_buffer = 300
        self.feature_history = 300
        self.body_strack_features = []      # (D:1285, D:1289: set once, never updated by the reference -- the update
        self.face_strack_features = []      #  loop's lists of the same name are locals; the tracks' features live on the device)
        self.proximity_thresh = 0.5
        self.appearance_thresh = 0.25
        self.buffer_size = int(frame_rate / 30.0 * self.track_buffer)
        self.max_time_lost = self.buffer_size
        self.kalman_filter = KalmanFilter()
        self.detector = object_detection_model
        self.body_encoder = body_feature_extractor_model
        self.face_encoder = face_feature_extractor_model
        if feat_dim is None:
            feat_dim = getattr(body_feature_extractor_model, "feature_size", 2048)
        if face_dim is None:
            face_dim = 0
        self.feat_dim, self.face_dim = int(feat_dim), int(face_dim)
        self.device_tracker = DeviceTracker(max_tracks, max_dets, self.feat_dim, self.face_dim, frame_rate,
                                            record_costs=record_costs, device=device, **tracker_options)
        self._mirror = {}          # track_id -> STrack
        self._last_dets = None

    # ---- tracker-only entry point
    def update_arrays(self, boxes, scores, body_feats, face_feats=None, bodies=None, export: bool = True):
        """boxes[M,4] integer tlbr pixels, scores[M], body_feats[M,D] float32 (L2-normalised),
        face_feats[M,Df] float32 or None; NumPy arrays or CUDA tensors.  NOTE: like the reference
        (quirk Q8) rows of the feature arrays of used detections are renormalised in place -- on
        the device copy.  `export=False` enqueues the frame and returns None (nothing is read back; the lists of this
        object are not refreshed); `removed_stracks` is the reference's list entry for entry as long as every frame is
        exported -- tracks that were removed during a run of un-exported frames are not recovered afterwards (their table
        slots may have been recycled)."""
        import torch

        def dev(a, dtype):
            if isinstance(a, torch.Tensor):
                return a.to(device="cuda", dtype=dtype).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype={torch.int32: np.int32, torch.float32: np.float32}[dtype])).cuda()

        m = len(boxes)
        b = dev(np.asarray(boxes).reshape(m, 4) if not isinstance(boxes, torch.Tensor) else boxes, torch.int32)
        s = dev(scores, torch.float32)
        f = dev(np.asarray(body_feats).reshape(m, self.feat_dim) if not isinstance(body_feats, torch.Tensor) else body_feats,
                torch.float32)
        ff = None
        if self.face_dim:
            if face_feats is None:
                raise ValueError("face_feats is required when the tracker was built with face_dim > 0")
            ff = dev(face_feats, torch.float32)
        self.device_tracker.step(b, s, f, ff)
        self.frame_id += 1
        self._keep = (b, s, f, ff)      # inputs stay alive until the stream has consumed them
        if not export:
            return None
        return self._export(bodies)

    def apply_gmc(self, H):
        """Camera-motion compensation of every tracked / lost track (STrack.multi_gmc, D:538-554)."""
        self.device_tracker.gmc(H)

    def _export(self, bodies=None):
        """Mirror the device lists: one export call per frame (both lists, pinned memory, one synchronisation); the lists
        handed out are TrackLists that build STrack mirrors on access.  Raises BtsError when the device flagged a truncated
        frame (track table full, solver edge list overflow): bts_tracker_export_lists turns the sticky flag into a status.
        With export=False nothing is read back -- poll `device_tracker.counts()['errors']`."""
        prev = self._snap
        snap = self.device_tracker.export_lists(self._epoch + 1)
        self._epoch += 1
        self._snap = snap
        # Who left both lists this frame?  total = previous total + births - vanished, births = ids handed out (D:565);
        # the exact set difference (and the bulk read of the table's id / state columns) only runs on frames where the
        # counters say somebody did
        if prev is not None:
            births = snap.counts["last_id"] - prev.counts["last_id"]
            if births < 0 or prev.n(0) + prev.n(1) + births != snap.n(0) + snap.n(1):
                self._finalise_vanished(prev, snap)
        self.tracked_stracks = TrackList(self, snap, 0)
        self.lost_stracks = TrackList(self, snap, 1)
        # removed_stracks.extend(removed_stracks) of D:1635 happens in the frame that marks a track removed; an expired lost
        # track stays on lost_stracks for that one frame (it is filtered against the PREVIOUS removals, Q6): it shows up here
        # as a lost row in state Removed.  The frame after, the expiry loop (D:1623-1627) runs over it once more -- it is
        # appended a SECOND time -- and only then does it leave the list: _finalise_vanished above parked those repeats in
        # self._repeats; both kinds are appended in the order of the loop, i.e. of the previous frame's lost list.
        # (Unconfirmed tracks that stage 3 left unmatched leave the lists at once: they were appended by _finalise_vanished
        # -- the reference appends them first too.)
        lost_ints = snap.ints[1]
        fresh = []
        if len(lost_ints):
            for k in np.nonzero(lost_ints[:, 1] == TrackState.Removed)[0].tolist():
                tid = int(lost_ints[k, 0])
                if tid not in self._removed_live:
                    self._removed_live.add(tid)
                    fresh.append((tid, self._mirror_at(snap, 1, k)))
        repeats, self._repeats = self._repeats, []
        if repeats and fresh:
            order = {int(t): i for i, t in enumerate(prev.ints[1][:, 0].tolist())}
            both = sorted(repeats + fresh, key=lambda e: order.get(e[0], len(order)))
        else:
            both = repeats or fresh
        self.removed_stracks.extend(st for _, st in both)
        if self._removed_live:
            # a removed track that is matched again before it is filtered out is tracked again (the reference's "zombie":
            # the same object stays on removed_stracks, and may be appended once more when it is removed again)
            for which in (0, 1):
                ints = snap.ints[which]
                if len(ints):
                    back = ints[(ints[:, 1] != TrackState.Removed) & np.isin(ints[:, 0], list(self._removed_live)), 0]
                    self._removed_live.difference_update(back.tolist())
        if bodies is not None:
            self._attach_bodies(snap, bodies)
        if self.lazy_lists is False or (self.lazy_lists is None and self._plain_call):
            self.tracked_stracks, self.lost_stracks = list(self.tracked_stracks), list(self.lost_stracks)
        return self.tracked_stracks

    def _mirror_at(self, snap, which, k):
        """The STrack mirror of entry k of a list (TrackList access): the same object for the same track across frames."""
        tid = snap.ids(which)[k]
        st = self._mirror.get(tid)
        if st is None:
            st = STrack._new_mirror(self, tid)
            if snap.epoch == self._epoch:
                self._mirror[tid] = st
            else:                       # an entry of a list kept from an earlier frame whose track is gone: a frozen record
                st._bind(snap, which, k)
                st.__dict__["_live"] = False
                return st
        if snap.epoch == self._epoch and st.__dict__["_epoch"] != snap.epoch:
            st._bind(snap, which, k)
        return st

    def _refresh(self, st):
        """First access after a frame to a mirror the caller kept: re-read its row of the current snapshot."""
        loc = self._snap.locate(st.track_id) if self._snap is not None else None
        if loc is None:
            st.__dict__["_epoch"] = self._epoch
        else:
            st._bind(self._snap, *loc)

    def _finalise_vanished(self, prev, snap):
        """Tracks of the previous frame's lists that are on neither list now: removed (D:1609-1612, D:1623-1627: they go to
        removed_stracks) or dropped by remove_duplicate_stracks (D:1637: they just leave the lists).  The table slot still
        holds the track this frame, so ONE bulk read of the id / state columns tells which; a mirror keeps its last values
        (and, with keep_removed_state, its covariance / feature rows) because the slot may be recycled from the next frame on."""
        prev_ints = np.concatenate([prev.ints[0], prev.ints[1]]) if prev.n(0) + prev.n(1) else np.zeros((0, 8), np.int32)
        cur_ids = np.concatenate([snap.ints[0][:, 0], snap.ints[1][:, 0]])
        gone = np.nonzero(~np.isin(prev_ints[:, 0], cur_ids))[0]
        if len(gone) == 0:
            return
        dt = self.device_tracker
        dev_id, dev_state = dt.read_column(0), dt.read_column(1)
        n0 = prev.n(0)
        for r in gone.tolist():
            tid, slot = int(prev_ints[r, 0]), int(prev_ints[r, 7])
            removed = dev_id[slot] == tid and dev_state[slot] == TrackState.Removed
            st = self._mirror.pop(tid, None)
            if st is None:
                if not removed:
                    self._removed_live.discard(tid)
                    continue                # never handed out and not removed: nobody can refer to it
                st = STrack._new_mirror(self, tid)
            st._bind(prev, 0 if r < n0 else 1, r if r < n0 else r - n0)
            d = st.__dict__
            d["_epoch"] = self._epoch
            if self.keep_removed_state and dev_id[slot] == tid:
                st.covariance, st.body_curr_feature, st.body_smooth_feature           # (fetched while the slot still holds the track)
                if self.face_dim:
                    st.face_curr_feature, st.face_smooth_feature
            d["_live"] = False
            if removed:
                d["state"] = TrackState.Removed
                self._removed_live.discard(tid)
                if r >= n0:       # off the lost list: expired last frame (appended then, and once more now) or expired and
                    self._repeats.append((tid, st))      # dropped as a duplicate in one frame -- _export puts it in loop order
                else:
                    self.removed_stracks.append(st)

    def _attach_bodies(self, snap, bodies):
        """update(image): the matched detection's Body goes onto its track and the track id onto the Body and its parts
        (D:583, D:609, D:1638) -- for the tracks matched or born this frame, the only ones whose body changed."""
        for which in (0, 1):
            ints = snap.ints[which]
            if len(ints) == 0:
                continue
            hit = np.nonzero((ints[:, 3] == self.frame_id) & (ints[:, 6] >= 0) & (ints[:, 6] < len(bodies)))[0]
            for k in hit.tolist():
                st = self._mirror_at(snap, which, k)
                st.body = bodies[int(ints[k, 6])]
                st.propagate_trackid_to_related_objects()

    # ---- reference signature (detector + encoders are the caller's objects, unchanged)
    def update(self, image: np.ndarray) -> List[STrack]:
        """D:1291: detect -> attach parts (D:1303-1411, on the device) -> encode -> track."""
        if self.detector is None or self.body_encoder is None:
            raise RuntimeError("update(image) needs detector/encoder models; use update_arrays for pre-extracted detections")
        boxes, _, _, _ = attach_parts(self.detector(image))
        m = len(boxes)
        arr = np.asarray([[b.x1, b.y1, b.x2, b.y2] for b in boxes], dtype=np.int32).reshape(m, 4)
        scores = np.asarray([b.score for b in boxes], dtype=np.float32)
        feats = np.zeros((m, self.feat_dim), dtype=np.float32)
        ffeats = None
        if m:
            crops = [image[b.y1:b.y2, b.x1:b.x2, :] for b in boxes]
            # the encoder's own similarity output is not used: the device computes it (D:1599 form)
            _, feats = self.body_encoder(base_images=crops, target_features=np.zeros((0, self.feat_dim), dtype=np.float32))
            if self.face_dim and self.face_encoder is not None:
                blank = getattr(self.face_encoder, "blank_image", None)                  # D:1440-1444
                fcrops = [image[b.head.face.y1:b.head.face.y2, b.head.face.x1:b.head.face.x2, :]
                          if b.head is not None and b.head.face is not None else
                          (blank if blank is not None else np.zeros((1, 1, 3), dtype=np.float32)) for b in boxes]
                ffeats, _ = self.face_encoder(base_images=fcrops,        # features first: D:1478-1480
                                               target_features=np.zeros((0, self.face_dim), dtype=np.float32))
        if self.face_dim and ffeats is None:
            ffeats = np.zeros((m, self.face_dim), dtype=np.float32)
        self._plain_call = True          # (the reference's signature returns the reference's type: plain lists)
        try:
            return self.update_arrays(arr, scores, np.asarray(feats, dtype=np.float32), ffeats, bodies=boxes)
        finally:
            self._plain_call = False


# list helpers with the reference's semantics (D:1642-1680), on host mirrors
def joint_stracks(tlista, tlistb):
    exists, res = {}, []
    for t in tlista:
        exists[t.track_id] = 1
        res.append(t)
    for t in tlistb:
        if not exists.get(t.track_id, 0):
            exists[t.track_id] = 1
            res.append(t)
    return res


def sub_stracks(tlista, tlistb):
    stracks = {t.track_id: t for t in tlista}
    for t in tlistb:
        if t.track_id in stracks:
            del stracks[t.track_id]
    return list(stracks.values())


def remove_duplicate_stracks(stracksa, stracksb):
    from .matching import dedupe_flags
    da, db = dedupe_flags(stracksa, stracksb, [t.frame_id - t.start_frame for t in stracksa],
                          [t.frame_id - t.start_frame for t in stracksb], 0.15)
    return ([t for i, t in enumerate(stracksa) if not da[i]], [t for i, t in enumerate(stracksb) if not db[i]])
