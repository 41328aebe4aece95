"""Drive the *reference module itself* with pre-extracted arrays (build container only).

/root/reference is read-only and does not exist on the GPU box: nothing in the test suite
imports this file except tests/golden/make_golden.py and tests that are skipped when the
reference tree is absent.  The reference is imported unmodified; `cv2` and `lap` (both absent
from this image) are pre-seeded in sys.modules -- `lap.lapjv` is served by the oracle's
restatement (oracle/lapjv_oracle.c), which is why lapjv tie-breaking stays "parity unpinned".

The detector / ReID encoders are out of scope: they are replaced by fakes that
replay the arrays of the synthetic stream through the exact call signatures BoTSORT.update uses
(D:1301, D:1453-1460, D:1473-1480).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF_DIR = "/root/reference"
REF_FILE = os.path.join(REF_DIR, "demo_bottrack_onnx_tflite.py")
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


def reference_available() -> bool:
    return os.path.exists(REF_FILE)


def load_reference():
    """Import the reference module with cv2/lap stubbed; returns the module object."""
    from oracle import bts_oracle
    if "demo_bottrack_onnx_tflite" in sys.modules:
        return sys.modules["demo_bottrack_onnx_tflite"]
    sys.dont_write_bytecode = True
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    lap = types.ModuleType("lap")
    lap.lapjv = bts_oracle.lapjv
    sys.modules["lap"] = lap
    sys.path.insert(0, REF_DIR)
    try:
        import demo_bottrack_onnx_tflite as mod   # noqa
    finally:
        sys.path.remove(REF_DIR)
    return mod


def _cos(t, d):
    return np.maximum(0.0, np.matmul(np.asarray(t, dtype=np.float32),
                                     np.asarray(d, dtype=np.float32).transpose(1, 0)))


class FakeDetector:
    def __init__(self, mod):
        self.mod = mod
        self.boxes = None
        self.scores = None

    def __call__(self, image):
        out = []
        for b, s in zip(self.boxes, self.scores):
            x1, y1, x2, y2 = (int(v) for v in b)
            out.append(self.mod.Box(trackid=0, classid=0, score=float(s), x1=x1, y1=y1, x2=x2, y2=y2,
                                    cx=(x1 + x2) // 2, cy=(y1 + y2) // 2, is_used=False))
        return out


class FakeBodyEncoder:
    """(similarities[Ndet,Ntrk], features[Ndet,D]) -- D:1453-1460."""

    def __init__(self, dim):
        self.feature_size = dim
        self.feats = None
        self.sim_override = None       # optional callable (sims[N,M]) -> sims[N,M]

    def __call__(self, *, base_images, target_features):
        t = np.asarray(target_features, dtype=np.float32).reshape(-1, self.feature_size)
        sims = _cos(t, self.feats)                      # [Ntrk, Ndet]
        if self.sim_override is not None:
            sims = self.sim_override(sims)
        return sims.transpose(1, 0), self.feats


class FakeFaceEncoder:
    """Returns (features, similarities) -- swapped order, D:1478-1480."""

    def __init__(self, dim):
        self.feature_size = dim
        self._input_shapes = [["N", 3, 8, 8]]
        self.feats = None
        self.sim_override = None

    def __call__(self, *, base_images, target_features):
        n_det = len(base_images)
        if self.feats is None:
            t = np.asarray(target_features, dtype=np.float32).reshape(-1, self.feature_size)
            return (np.zeros((n_det, self.feature_size), dtype=np.float32),
                    np.zeros((n_det, len(t)), dtype=np.float32))
        t = np.asarray(target_features, dtype=np.float32).reshape(-1, self.feature_size)
        sims = _cos(t, self.feats)
        if self.sim_override is not None:
            sims = self.sim_override(sims)
        return self.feats, sims.transpose(1, 0)


def table(tracks) -> dict:
    return {
        "track_id": np.asarray([t.track_id for t in tracks], dtype=np.int64),
        "state": np.asarray([t.state for t in tracks], dtype=np.int64),
        "is_activated": np.asarray([int(t.is_activated) for t in tracks], dtype=np.int64),
        "frame_id": np.asarray([t.frame_id for t in tracks], dtype=np.int64),
        "start_frame": np.asarray([t.start_frame for t in tracks], dtype=np.int64),
        "tracklet_len": np.asarray([t.tracklet_len for t in tracks], dtype=np.int64),
        "mean": np.asarray([np.asarray(t.mean, dtype=np.float64) for t in tracks]).reshape(-1, 8),
    }


class ReferenceRunner:
    """One reference BoTSORT driven by arrays; records the three assignment calls per frame."""

    def __init__(self, feat_dim, face_dim=0, frame_rate=30):
        self.mod = load_reference()
        self.det = FakeDetector(self.mod)
        self.body = FakeBodyEncoder(feat_dim)
        self.face = FakeFaceEncoder(face_dim if face_dim else 8)
        self.use_face = bool(face_dim)
        self.bot = self.mod.BoTSORT(self.det, self.body, self.face, frame_rate=frame_rate)
        self.calls = []
        self.iou_calls = []

    def update_arrays(self, boxes, scores, feats, face_feats=None):
        mod = self.mod
        self.det.boxes, self.det.scores = boxes, scores
        self.body.feats = feats
        self.face.feats = face_feats if self.use_face else None
        self.calls = []
        self.iou_calls = []
        orig_la, orig_iou = mod.linear_assignment, mod.iou_distance

        def la(cost, thresh):
            res = orig_la(cost, thresh)
            self.calls.append((np.array(cost, dtype=np.float64, copy=True), thresh, res))
            return res

        def iou(a, b):
            r = orig_iou(a, b)
            self.iou_calls.append(np.array(r, copy=True))
            return r

        mod.linear_assignment, mod.iou_distance = la, iou
        try:
            with np.errstate(invalid="ignore", divide="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out = self.bot.update(np.zeros((1, 1, 3), dtype=np.uint8))
        finally:
            mod.linear_assignment, mod.iou_distance = orig_la, orig_iou
        return out

    def stage(self, k):
        cost, thresh, (m, ua, ub) = self.calls[k]
        return {
            "cost": cost,
            "m": np.asarray(m, dtype=np.int64).reshape(-1, 2),
            "u_trk": np.asarray(ua, dtype=np.int64),
            "u_det": np.asarray(ub, dtype=np.int64),
        }
