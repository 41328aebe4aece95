"""Seeded synthetic detection streams for the tracker-only benchmark and the parity tests.

The tracker hot path takes *pre-extracted* arrays (BASELINE.json north_star): per frame
integer-pixel boxes, detector scores and L2-normalised ReID features.  This generator is
shared by the golden-fixture script, the CPU oracle, the GPU parity tests and bench.py so that
every leg sees byte-identical inputs for a given (config, seed).

Scene model (with the feature noise scaled by 1/sqrt(D) so the appearance
term of the fused cost is actually exercised at every D):
  * K identities on a square canvas of side ceil(125*sqrt(K)) px, w~U(30,80), h~U(60,160),
    centres do a N(0, 2 px) random walk per frame;
  * frame 0 shows every identity with score ~U(0.9,1) (all become confirmed tracks, since
    tracks born at frame 1 are activated immediately, reference D:565-566);
  * later frames show M identities drawn without replacement; 1/8 of them get a low score
    ~U(0.12,0.38) (second association), the rest ~U(0.9,1); optional churn replaces a fraction
    of the detections with brand-new identities (births / unconfirmed-track stage);
  * detection feature = normalise(prototype + (0.3/sqrt(D)) * N(0,1)), float32.

Adversarial variants (bench.py --workload crowd | hard-reid, tests): `side_scale` shrinks the canvas (40 instead of
125 px per sqrt(identity): boxes overlap 5-10 deep, the IoU graph gets large connected components); `clusters` draws
the prototypes around that many shared centres with per-identity spreads so that impostor cosines of 0.5-0.8 are
commonplace (what post-ReLU ReID embeddings look like), instead of the near-orthogonal i.i.d. prototypes; `dup_frac`
(workload `dup-box`) turns a fraction of the detections into exact copies of another box with unrelated embeddings -- exact
cost ties, the only situation in which the fast assignment solver may differ from lap.lapjv.  `clusters` draws
its extra randomness from a generator of its own and the reduced canvas margin applies only when `side_scale != 125`, so the
streams of the plain configurations are bit-identical to round 1's (tests/test_fixtures_regenerate.py regenerates the fixtures
and compares).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

# name -> (tracks K, detections M, feature dim D): BASELINE.json configs + the headline target
CONFIGS = {
    "c1": (64, 32, 2048),
    "c2": (512, 256, 2048),
    "c3": (4096, 1024, 2048),
    "target": (10000, 2000, 2048),
    "c5": (16384, 4096, 384),
}


@dataclass
class Frame:
    boxes: np.ndarray            # int32 [M,4] tlbr pixels
    scores: np.ndarray           # float32 [M]
    feats: np.ndarray            # float32 [M,D], L2-normalised
    face_feats: Optional[np.ndarray]   # float32 [M,Df] or None
    ident: np.ndarray            # int64 [M] ground-truth identity (not seen by the tracker)


class SyntheticStream:
    def __init__(self, n_ids: int, n_dets: int, feat_dim: int, seed: int = 1234,
                 churn: float = 0.0, low_frac: float = 0.125, face_dim: int = 0,
                 feat_noise: float = 0.3, side_scale: float = 125.0, clusters: int = 0, dup_frac: float = 0.0):
        self.rng = np.random.default_rng(seed)
        self.K0 = n_ids
        self.M = n_dets
        self.D = feat_dim
        self.Df = face_dim
        self.churn = churn
        self.low_frac = low_frac
        self.feat_noise = feat_noise
        self.side_scale = float(side_scale)
        self.side = int(math.ceil(side_scale * math.sqrt(n_ids)))
        self.clusters = int(clusters)
        self.dup_frac = float(dup_frac)
        self._dr = np.random.default_rng([int(seed), 0xd0b1]) if dup_frac else None     # (never touches self.rng)
        self._xr = np.random.default_rng([int(seed), 0x5eed]) if clusters else None      # (never touches self.rng)
        self._centres = None
        self.cx = np.empty(0)
        self.cy = np.empty(0)
        self.w = np.empty(0)
        self.h = np.empty(0)
        self.proto = np.empty((0, feat_dim), dtype=np.float32)
        self.fproto = np.empty((0, face_dim), dtype=np.float32)
        self._spawn(n_ids)
        self.t = 0

    def _spawn(self, k: int):
        r = self.rng
        # 100 px as in every fixture of rounds 1-2; only a shrunken canvas (the crowd scene, side_scale != 125) takes a
        # margin that still leaves room when the canvas is small (side/8 < 100 px also happens on PLAIN streams with
        # fewer than 41 identities: they keep the 100 px their fixtures were recorded with)
        margin = 100.0 if self.side_scale == 125.0 else min(100.0, self.side / 8.0)
        self.cx = np.concatenate([self.cx, r.uniform(margin, self.side - margin, k)])
        self.cy = np.concatenate([self.cy, r.uniform(margin, self.side - margin, k)])
        self.w = np.concatenate([self.w, r.uniform(30.0, 80.0, k)])
        self.h = np.concatenate([self.h, r.uniform(60.0, 160.0, k)])
        p = r.standard_normal((k, self.D)).astype(np.float32)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        if self.clusters:
            # prototype = normalise(centre + s * unit noise), s ~ U(0.5, 1): cos(i, j) = 1 / sqrt((1 + s_i^2)(1 + s_j^2))
            # in [0.5, 0.8] for two identities of one cluster
            x = self._xr
            if self._centres is None:
                c = x.standard_normal((self.clusters, self.D)).astype(np.float32)
                self._centres = c / np.linalg.norm(c, axis=1, keepdims=True)
            which = x.integers(0, self.clusters, k)
            spread = x.uniform(0.5, 1.0, (k, 1)).astype(np.float32)
            p = self._centres[which] + spread * p
            p /= np.linalg.norm(p, axis=1, keepdims=True)
        self.proto = np.concatenate([self.proto, p.astype(np.float32)])
        if self.Df:
            q = r.standard_normal((k, self.Df)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            self.fproto = np.concatenate([self.fproto, q])

    def _observe(self, ids: np.ndarray, scores: np.ndarray) -> Frame:
        r = self.rng
        x1 = np.rint(self.cx[ids] - self.w[ids] / 2)
        y1 = np.rint(self.cy[ids] - self.h[ids] / 2)
        x2 = np.rint(self.cx[ids] + self.w[ids] / 2)
        y2 = np.rint(self.cy[ids] + self.h[ids] / 2)
        boxes = np.stack([x1, y1, x2, y2], axis=1).astype(np.int32)
        f = self.proto[ids] + (self.feat_noise / math.sqrt(self.D)) * \
            r.standard_normal((len(ids), self.D)).astype(np.float32)
        f = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
        ff = None
        if self.Df:
            ff = self.fproto[ids] + (self.feat_noise / math.sqrt(self.Df)) * \
                r.standard_normal((len(ids), self.Df)).astype(np.float32)
            ff = (ff / np.linalg.norm(ff, axis=1, keepdims=True)).astype(np.float32)
        return Frame(boxes, scores.astype(np.float32), np.ascontiguousarray(f), ff, ids.astype(np.int64))

    def next(self) -> Frame:
        r = self.rng
        if self.t == 0:
            ids = np.arange(self.K0)
            fr = self._observe(ids, r.uniform(0.9, 1.0, len(ids)))
            self.t += 1
            return fr
        k = len(self.cx)
        self.cx += r.normal(0.0, 2.0, k)
        self.cy += r.normal(0.0, 2.0, k)
        n_new = int(round(self.churn * self.M))
        ids = r.choice(k, size=self.M - n_new, replace=False)
        if n_new:
            self._spawn(n_new)
            ids = np.concatenate([ids, np.arange(k, k + n_new)])
        n_low = int(self.low_frac * self.M)
        scores = r.uniform(0.9, 1.0, len(ids))
        if n_low:
            low_pos = r.choice(len(ids) - n_new, size=min(n_low, len(ids) - n_new - 1), replace=False) \
                if len(ids) - n_new > 1 else np.empty(0, dtype=np.int64)
            scores[low_pos] = r.uniform(0.12, 0.38, len(low_pos))
        perm = r.permutation(len(ids))
        fr = self._observe(ids[perm], scores[perm])
        if self.dup_frac:
            self._duplicate(fr)
        self.t += 1
        return fr

    def _duplicate(self, fr: "Frame"):
        """The `dup-box` scene: a fraction of the frame's detections become EXACT copies of another detection's box, and both
        copies get unrelated random embeddings (appearance gated to 1 for every track): the two columns of the cost matrix
        are identical IoU costs -- exact ties, the case in which lap.lapjv's answer depends on its scan orders (DESIGN.md
        section 4).  What a detector emits when NMS is off or class-wise (same box, two classes)."""
        x = self._dr
        m = len(fr.boxes)
        k = min(int(round(self.dup_frac * m)), m // 2)
        if k == 0:
            return
        hi = np.nonzero(fr.scores > 0.5)[0]
        if len(hi) < 2 * k:
            return
        pick = x.choice(hi, size=2 * k, replace=False)
        src, dst = pick[:k], pick[k:]
        fr.boxes[dst] = fr.boxes[src]
        both = np.concatenate([src, dst])
        f = x.standard_normal((2 * k, self.D)).astype(np.float32)
        fr.feats[both] = f / np.linalg.norm(f, axis=1, keepdims=True)


def make_stream(config: str, seed: int = 1234, **kw) -> SyntheticStream:
    k, m, d = CONFIGS[config]
    return SyntheticStream(k, m, d, seed=seed, **kw)


# bench / test workloads: the BASELINE configurations plus the adversarial scenes.  "name@config" runs an adversarial
# scene at another configuration's size (e.g. crowd@target).
ADVERSARIAL = {
    "crowd": dict(side_scale=40.0),          # boxes overlap 5-10 deep
    "hard-reid": dict(clusters=32),          # impostor cosines 0.5-0.8
    "crowd-hard-reid": dict(side_scale=40.0, clusters=32),
    "dup-box": dict(dup_frac=0.05),           # 5 % of the detections are exact copies of another box: cost ties
}


def _split(name: str):
    base, _, at = name.partition("@")
    if base in ADVERSARIAL:
        return base, CONFIGS[at or "c3"]
    return None, CONFIGS[base]


def workload_dims(name: str):
    return _split(name)[1]


def workload_text(name: str) -> str:
    adv, (k, m, d) = _split(name)
    txt = "%d tracks x %d detections/frame x %d-d ReID" % (k, m, d)
    if adv == "crowd" or adv == "crowd-hard-reid":
        txt += ", dense crowd (canvas 40 px per sqrt(identity): boxes overlap 5-10 deep)"
    if adv == "hard-reid" or adv == "crowd-hard-reid":
        txt += ", clustered embeddings (32 centres, impostor cosines 0.5-0.8)"
    if adv == "dup-box":
        txt += ", 5 % of the detections exact copies of another box with unrelated embeddings (exact cost ties)"
    return txt


def make_workload(name: str, seed: int = 1234, **kw) -> SyntheticStream:
    adv, (k, m, d) = _split(name)
    opts = dict(ADVERSARIAL[adv]) if adv else {}
    opts.update(kw)
    return SyntheticStream(k, m, d, seed=seed, **opts)
