"""
oracle/bts_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (NumPy/SciPy + a small C helper) restatement of the BoT-SORT tracker hot path of
PINTO0309/BoT-SORT-ONNX-TensorRT, file `demo_bottrack_onnx_tflite.py` (cited below as D:line).
It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it.  The product package never does.

What is restated, and from where:
  * Kalman filter (8-state constant velocity, 4x4 innovation Cholesky)        D:118-380
  * track record + lifecycle (activate / re_activate / update / mark_*)      D:382-437, D:439-688
  * association cascade + list merging of BoTSORT.update (tracking half)      D:1414-1639
  * list helpers joint / sub / remove_duplicate                               D:1642-1680
  * linear_assignment -> lap.lapjv(extend_cost=True, cost_limit=thresh)       D:1682-1693
    (lap==0.4.0 is a third-party dependency whose source is absent: its published
     algorithm is restated in oracle/lapjv_oracle.c -- tie-breaking "parity unpinned")
  * bbox_iou / bbox_ious / iou_distance                                       D:1695-1761
  * fused costs of stage 1 and stage 3                                        D:1539-1554, D:1590-1602

Pinning: tests/golden/make_golden.py imports the *reference module itself* in the build
container (cv2/lap stubbed, detector/encoders replaced by array-replaying fakes), records its
outputs, and tests/test_oracle_vs_golden.py checks this file against those vectors.  The
reference runs under this image's NumPy 2.2 / SciPy 1.15, so its NEP-50 scalar-promotion
behaviour (python-float x np.float32 stays float32) is what is pinned; the dtype flow is
reproduced deliberately (see `iou_cost`, `KalmanOracle.initiate`, `KalmanOracle.project`).

The ReID "similarity" matrices are produced inside ONNX graphs in the reference (D:1453-1483);
with pre-extracted features the only in-repo definition is D:1599, `max(0, A @ B.T)`, which is
adopted for every stage here (and by the fake encoders used to drive the reference).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# Track states, D:382-387
NEW, TRACKED, LOST, LONGLOST, REMOVED = 0, 1, 2, 3, 4


# ----------------------------------------------------------------------------------------------
# linear assignment (A18)
# ----------------------------------------------------------------------------------------------
def _load_lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_lapjv.so")
        src = os.path.join(_HERE, "lapjv_oracle.c")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        lib = ctypes.CDLL(so)
        lib.bts_oracle_lapjv_extended.restype = ctypes.c_int
        lib.bts_oracle_lapjv_extended.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.bts_oracle_lapjv_square.restype = ctypes.c_int
        lib.bts_oracle_lapjv_square.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def lapjv(cost: np.ndarray, extend_cost: bool = True, cost_limit: float = np.inf):
    """Stand-in for `lap.lapjv` (third-party, lap==0.4.0) as called at D:1686.

    Returns (opt, x, y): x[i] = column of row i or -1, y[j] = row of column j or -1.
    Only the `extend_cost=True, cost_limit<inf` mode the reference uses is provided.
    """
    if not extend_cost or not np.isfinite(cost_limit):
        raise NotImplementedError("only lapjv(extend_cost=True, cost_limit=<finite>) is restated")
    c = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = c.shape
    x = np.empty(n, dtype=np.int32)
    y = np.empty(m, dtype=np.int32)
    opt = ctypes.c_double(0.0)
    rc = _load_lib().bts_oracle_lapjv_extended(
        n, m, c.ctypes.data, float(cost_limit), x.ctypes.data, y.ctypes.data, ctypes.byref(opt))
    if rc != 0:
        raise MemoryError("oracle lapjv allocation failed")
    return opt.value, x, y


def linear_assignment(cost: np.ndarray, thresh: float):
    """D:1682-1693.  Return types follow the reference (Q14): tuples for an empty cost,
    ndarrays otherwise; `matches` has shape (0,) when nothing matched."""
    if cost.size == 0:
        return (np.empty((0, 2), dtype=int), tuple(range(cost.shape[0])),
                tuple(range(cost.shape[1])))
    _, x, y = lapjv(cost, extend_cost=True, cost_limit=thresh)
    rows = np.nonzero(x >= 0)[0]
    matches = np.asarray([[int(r), int(x[r])] for r in rows])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]


# ----------------------------------------------------------------------------------------------
# IoU cost (A13)
# ----------------------------------------------------------------------------------------------
def iou_pair(a, b):
    """One pair, written with Python scalars exactly like D:1695-1713 evaluates: the dtype of
    every intermediate follows NumPy scalar arithmetic (this is what makes mixed f64/f32 inputs
    come out the way the reference computes them)."""
    x_lo = b[0] if b[0] > a[0] else a[0]
    y_lo = b[1] if b[1] > a[1] else a[1]
    x_hi = b[2] if b[2] < a[2] else a[2]
    y_hi = b[3] if b[3] < a[3] else a[3]
    if x_hi <= x_lo or y_hi <= y_lo:
        return 0.0
    inter = (x_hi - x_lo) * (y_hi - y_lo)
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    return inter / float(area_a + area_b - inter)


def iou_cost_pairwise(a_boxes: Sequence[np.ndarray], b_boxes: Sequence[np.ndarray]) -> np.ndarray:
    """Structure-faithful N*M Python loop (D:1731-1761): the CPU baseline's hot loop."""
    ious = np.zeros((len(a_boxes), len(b_boxes)), dtype=np.float32)
    if ious.size == 0:
        return 1 - ious
    ious = np.array([[iou_pair(a, b) for b in b_boxes] for a in a_boxes])
    return 1 - ious


def _as_box_array(boxes: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Stack boxes into f64 [K,4] plus a per-box 'is float32' flag."""
    k = len(boxes)
    out = np.empty((k, 4), dtype=np.float64)
    f32 = np.zeros(k, dtype=bool)
    for i, b in enumerate(boxes):
        out[i] = b                      # f32 -> f64 is exact
        f32[i] = (np.asarray(b).dtype == np.float32)
    return out, f32


def iou_cost_arrays(a: np.ndarray, a_f32: np.ndarray, b: np.ndarray, b_f32: np.ndarray) -> np.ndarray:
    """Vectorised equivalent of `iou_cost_pairwise` with the same per-element dtype flow.

    a[N,4], b[M,4] are float64 holders; a_f32[N] / b_f32[M] say whether the box was float32 in
    the reference.  An intermediate is float32 exactly when both of its operands are float32
    scalars (max/min return one of their arguments, ties return the first = `a`); the final
    division is float32 when the intersection area is float32 (NEP 50: np.float32 / python
    float stays float32).  One deliberate deviation, documented in DESIGN.md: when *every*
    entry of the matrix is a float32 scalar NumPy builds a float32 matrix and evaluates
    `1 - iou` in float32; this function always evaluates `1 - iou` in float64 (|diff| <= 6e-8).
    """
    n, m = a.shape[0], b.shape[0]
    if n == 0 or m == 0:
        return 1 - np.zeros((n, m), dtype=np.float32)
    A = a[:, None, :]
    B = b[None, :, :]
    af = a_f32[:, None]
    bf = b_f32[None, :]
    f32 = np.float32

    def pick_lo(k):
        take_b = B[..., k] > A[..., k]
        return np.where(take_b, B[..., k], A[..., k]), np.where(take_b, bf, af)

    def pick_hi(k):
        take_b = B[..., k] < A[..., k]
        return np.where(take_b, B[..., k], A[..., k]), np.where(take_b, bf, af)

    x_lo, x_lo32 = pick_lo(0)
    y_lo, y_lo32 = pick_lo(1)
    x_hi, x_hi32 = pick_hi(2)
    y_hi, y_hi32 = pick_hi(3)
    empty = (x_hi <= x_lo) | (y_hi <= y_lo)

    def sub(hi, lo, both32):
        d64 = hi - lo
        d32 = (hi.astype(f32) - lo.astype(f32)).astype(np.float64)
        return np.where(both32, d32, d64)

    def mul(p, q, both32):
        r64 = p * q
        r32 = (p.astype(f32) * q.astype(f32)).astype(np.float64)
        return np.where(both32, r32, r64)

    w32 = x_lo32 & x_hi32
    h32 = y_lo32 & y_hi32
    w = sub(x_hi, x_lo, w32)
    h = sub(y_hi, y_lo, h32)
    i32 = w32 & h32
    inter = mul(w, h, i32)

    def area(box, is32):
        bw = sub(box[..., 2], box[..., 0], is32)
        bh = sub(box[..., 3], box[..., 1], is32)
        return mul(bw, bh, is32)

    area_a = area(A, af)          # [N,1]
    area_b = area(B, bf)          # [1,M]
    ab32 = af & bf
    s64 = area_a + area_b
    s32 = (area_a.astype(f32) + area_b.astype(f32)).astype(np.float64)
    s = np.where(ab32, s32, s64)
    all32 = ab32 & i32            # i32 implies nothing about ab32 when only one side is f32
    den64 = s - inter
    den32 = (s.astype(f32) - inter.astype(f32)).astype(np.float64)
    den = np.where(ab32 & i32, den32, den64)
    del all32
    with np.errstate(divide="ignore", invalid="ignore"):
        q64 = inter / den
        q32 = (inter.astype(f32) / den.astype(f32)).astype(np.float64)
    iou = np.where(i32, q32, q64)
    iou = np.where(empty, 0.0, iou)
    return 1 - iou


def iou_cost(a_boxes: Sequence[np.ndarray], b_boxes: Sequence[np.ndarray], faithful: bool = False):
    """`iou_distance` (D:1745-1761) on lists of tlbr boxes (np arrays of dtype f32 or f64)."""
    if faithful:
        return iou_cost_pairwise(a_boxes, b_boxes)
    a, af = _as_box_array(a_boxes)
    b, bf = _as_box_array(b_boxes)
    return iou_cost_arrays(a, af, b, bf)


# ----------------------------------------------------------------------------------------------
# detection-side part association (D:1715-1729, D:1763-1791, D:1379-1411)
# ----------------------------------------------------------------------------------------------
def box_iou_int(a, b) -> float:
    """bbox_iou_by_box (D:1715-1729) on integer (x1, y1, x2, y2) boxes."""
    x_lo, y_lo = max(a[0], b[0]), max(a[1], b[1])
    x_hi, y_hi = min(a[2], b[2]), min(a[3], b[3])
    if x_hi <= x_lo or y_hi <= y_lo:
        return 0.0
    inter = (x_hi - x_lo) * (y_hi - y_lo)
    return inter / float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def associate_parts(base_boxes, base_centres, target_boxes, target_centres, used=None, picks: int = 1):
    """Greedy attachment of `target` parts to `base` objects in base order -- what
    find_most_relevant_object (D:1763-1791) does inside the loops of D:1379-1411: each base takes the
    unused target with the largest IoU (> 0), ties by the smaller centre distance, then by list
    order; `picks` = 2 reproduces the hand1 / hand2 double pick.  Returns (idx[nb, picks] (-1 = none), used)."""
    nb, nt = len(base_boxes), len(target_boxes)
    used = np.zeros(nt, dtype=bool) if used is None else np.asarray(used, dtype=bool).copy()
    out = np.full((nb, picks), -1, dtype=np.int64)
    for b in range(nb):
        for p in range(picks):
            best, best_iou, best_d = -1, 0.0, float("inf")
            for t in range(nt):
                if used[t]:
                    continue
                iou = box_iou_int([int(v) for v in base_boxes[b]], [int(v) for v in target_boxes[t]])
                d = (int(base_centres[b][0] - target_centres[t][0]) ** 2 + int(base_centres[b][1] - target_centres[t][1]) ** 2) ** 0.5
                if iou > best_iou:
                    best, best_iou, best_d = t, iou, d
                elif iou > 0.0 and iou == best_iou and d < best_d:
                    best, best_d = t, d
            if best >= 0:
                used[best] = True
            out[b, p] = best
    return out, used


# ----------------------------------------------------------------------------------------------
# fused costs (A14-A17)
# ----------------------------------------------------------------------------------------------
def cosine_similarity(track_feats: np.ndarray, det_feats: np.ndarray) -> np.ndarray:
    """max(0, T @ D.T) in float32 -- the D:1599 definition, used for every stage. [N,M]."""
    t = np.asarray(track_feats, dtype=np.float32)
    d = np.asarray(det_feats, dtype=np.float32)
    return np.maximum(0.0, np.matmul(t, d.transpose(1, 0)))


def face_similarity_fixup(face_sim: np.ndarray) -> np.ndarray:
    """D:1481-1483: similarities that are 'close to 0.9999999' are zeroed."""
    out = np.array(face_sim, dtype=np.float32, copy=True)
    out[np.isclose(out, 0.9999999, atol=1e-08, rtol=1e-08)] = 0.0
    return out


def fuse_cost_stage1(iou_d: np.ndarray, body_sim: np.ndarray, face_sim: np.ndarray,
                     proximity_thresh: float = 0.5, appearance_thresh: float = 0.25) -> np.ndarray:
    """D:1539-1554.  The proximity gate only re-sets entries the appearance gate already set
    (quirk Q2), so the result is min(iou_d, emb) with emb=1 where min(body,face) distance
    exceeds the appearance threshold."""
    emb = 1.0 - body_sim
    face = 1.0 - face_sim
    far = np.minimum(emb, face) > appearance_thresh
    emb[far] = 1.0
    emb[np.logical_and(far, iou_d > proximity_thresh)] = 1.0
    return np.minimum(iou_d, emb)


def fuse_cost_stage3(iou_d: np.ndarray, body_sim: np.ndarray,
                     proximity_thresh: float = 0.5, appearance_thresh: float = 0.25) -> np.ndarray:
    """D:1590-1602 (body_sim already max(0, U @ B.T))."""
    emb = 1.0 - body_sim
    emb[emb > appearance_thresh] = 1.0
    emb[iou_d > proximity_thresh] = 1.0
    return np.minimum(iou_d, emb)


# ----------------------------------------------------------------------------------------------
# Kalman filter (A1-A7)
# ----------------------------------------------------------------------------------------------
CHI2INV95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070,
             6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}       # D:139-149


class KalmanOracle:
    """8-state (cx, cy, w, h and their velocities) constant-velocity filter, D:118-380."""

    chi2inv95 = CHI2INV95

    def __init__(self):
        self.F = np.eye(8)
        self.F[:4, 4:] = np.eye(4)            # dt = 1, D:151-158
        self.H = np.eye(4, 8)
        self.w_pos = 1.0 / 20                 # D:163
        self.w_vel = 1.0 / 160                # D:164

    def initiate(self, z: np.ndarray):
        """D:166-197.  With a float32 measurement the mean and (under NumPy 2) the covariance
        stay float32: python-float * np.float32 scalars are float32."""
        mean = np.concatenate([z, np.zeros_like(z)])
        w, h = z[2], z[3]
        sd = [2 * self.w_pos * w, 2 * self.w_pos * h, 2 * self.w_pos * w, 2 * self.w_pos * h,
              10 * self.w_vel * w, 10 * self.w_vel * h, 10 * self.w_vel * w, 10 * self.w_vel * h]
        return mean, np.diag(np.square(sd))

    def _process_noise_diag(self, w, h):
        return np.square(np.array([self.w_pos * w, self.w_pos * h, self.w_pos * w, self.w_pos * h,
                                   self.w_vel * w, self.w_vel * h, self.w_vel * w, self.w_vel * h]))

    def predict(self, mean: np.ndarray, cov: np.ndarray):
        """D:199-234."""
        sd_p = [self.w_pos * mean[2], self.w_pos * mean[3], self.w_pos * mean[2], self.w_pos * mean[3]]
        sd_v = [self.w_vel * mean[2], self.w_vel * mean[3], self.w_vel * mean[2], self.w_vel * mean[3]]
        q = np.diag(np.square(np.r_[sd_p, sd_v]))
        new_mean = np.dot(mean, self.F.T)
        new_cov = np.linalg.multi_dot((self.F, cov, self.F.T)) + q
        return new_mean, new_cov

    def multi_predict(self, mean: np.ndarray, cov: np.ndarray):
        """D:265-302, batched over the leading axis."""
        sd = [self.w_pos * mean[:, 2], self.w_pos * mean[:, 3], self.w_pos * mean[:, 2], self.w_pos * mean[:, 3],
              self.w_vel * mean[:, 2], self.w_vel * mean[:, 3], self.w_vel * mean[:, 2], self.w_vel * mean[:, 3]]
        sq = np.square(np.r_[sd[:4], sd[4:]]).T                 # [N,8]
        q = np.zeros((len(mean), 8, 8), dtype=sq.dtype)
        idx = np.arange(8)
        q[:, idx, idx] = sq
        new_mean = np.dot(mean, self.F.T)
        left = np.dot(self.F, cov).transpose((1, 0, 2))
        new_cov = np.dot(left, self.F.T) + q
        return new_mean, new_cov

    def project(self, mean: np.ndarray, cov: np.ndarray):
        """D:236-263."""
        sd = [self.w_pos * mean[2], self.w_pos * mean[3], self.w_pos * mean[2], self.w_pos * mean[3]]
        r = np.diag(np.square(sd))
        z = np.dot(self.H, mean)
        s = np.linalg.multi_dot((self.H, cov, self.H.T))
        return z, s + r

    def update(self, mean: np.ndarray, cov: np.ndarray, z: np.ndarray):
        """D:304-336: Cholesky of the 4x4 innovation covariance, gain by two triangular solves."""
        zhat, s = self.project(mean, cov)
        fac = scipy.linalg.cho_factor(s, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve(fac, np.dot(cov, self.H.T).T, check_finite=False).T
        innov = z - zhat
        new_mean = mean + np.dot(innov, gain.T)
        new_cov = cov - np.linalg.multi_dot((gain, s, gain.T))
        return new_mean, new_cov

    @staticmethod
    def multi_gmc(mean: np.ndarray, cov: np.ndarray, H: np.ndarray):
        """STrack.multi_gmc (D:538-554; its call sites are commented out at D:1534-1536): camera-motion
        warp of a batch of states, x <- kron(I4, R) x (+ t on the centre), P <- R8 P R8^T."""
        R8 = np.kron(np.eye(4, dtype=float), H[:2, :2])
        t = H[:2, 2]
        out_m, out_c = [], []
        for m, c in zip(mean, cov):
            m2 = R8.dot(m)
            m2[:2] += t
            out_m.append(m2)
            out_c.append(R8.dot(c).dot(R8.transpose()))
        return np.asarray(out_m), np.asarray(out_c)

    def gating_distance(self, mean, cov, measurements, only_position=False, metric="maha"):
        """D:338-380 (never called by the reference's update loop; API parity only)."""
        zhat, s = self.project(mean, cov)
        if only_position:
            zhat, s = zhat[:2], s[:2, :2]
            measurements = measurements[:, :2]
        d = measurements - zhat
        if metric == "gaussian":
            return np.sum(d * d, axis=1)
        if metric == "maha":
            low = np.linalg.cholesky(s)
            y = scipy.linalg.solve_triangular(low, d.T, lower=True, check_finite=False)
            return np.sum(y * y, axis=0)
        raise ValueError("invalid distance metric")


# ----------------------------------------------------------------------------------------------
# track record + tracker (A8-A12, A19-A21)
# ----------------------------------------------------------------------------------------------
class TrackRec:
    """One track / detection record: the fields of STrack (D:439-490) the hot path touches."""
    __slots__ = ("tlwh0", "mean", "cov", "is_activated", "score", "tracklet_len", "alpha",
                 "body_curr", "body_smooth", "face_curr", "face_smooth", "track_id", "state",
                 "frame_id", "start_frame", "det_index", "kf")

    def __init__(self, tlwh, score, body_feat=None, face_feat=None, det_index=-1):
        self.tlwh0 = np.asarray(tlwh, dtype=np.float32)     # D:465
        self.mean = None
        self.cov = None
        self.is_activated = False
        self.score = score
        self.tracklet_len = 0
        self.alpha = 0.9                                    # D:473
        self.body_curr = self.body_smooth = None
        self.face_curr = self.face_smooth = None
        self.track_id = 0
        self.state = NEW
        self.frame_id = 0
        self.start_frame = 0
        self.det_index = det_index
        self.kf = None
        if body_feat is not None:
            self.push_body(body_feat)
        if face_feat is not None:
            self.push_face(face_feat)

    # feature EMA, D:492-514 (first call aliases + renormalises the caller's row, Q8)
    def push_body(self, f):
        self.body_curr = f
        if self.body_smooth is None:
            self.body_smooth = f
        else:
            self.body_smooth = self.alpha * self.body_smooth + (1 - self.alpha) * f
        self.body_smooth /= np.linalg.norm(self.body_smooth)

    def push_face(self, f):
        self.face_curr = f
        if self.face_smooth is None:
            self.face_smooth = f
        else:
            self.face_smooth = self.alpha * self.face_smooth + (1 - self.alpha) * f
        with np.errstate(invalid="ignore", divide="ignore"):
            self.face_smooth /= np.linalg.norm(self.face_smooth)

    @property
    def end_frame(self):
        return self.frame_id

    @property
    def tlwh(self):                       # D:624-633
        if self.mean is None:
            return self.tlwh0.copy()
        r = self.mean[:4].copy()
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):                       # D:635-642
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r

    @staticmethod
    def tlwh_to_xywh(tlwh):               # D:663-670
        r = np.asarray(tlwh).copy()
        r[:2] += r[2:] / 2
        return r

    def activate(self, kf: KalmanOracle, frame_id: int, tracker: "TrackerOracle"):   # D:556-568
        self.kf = kf
        self.track_id = tracker._next_id()
        self.mean, self.cov = kf.initiate(self.tlwh_to_xywh(self.tlwh0))
        self.tracklet_len = 0
        self.state = TRACKED
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = frame_id
        self.start_frame = frame_id

    def _absorb(self, det: "TrackRec"):
        self.mean, self.cov = self.kf.update(self.mean, self.cov, self.tlwh_to_xywh(det.tlwh))
        if det.body_curr is not None:
            self.push_body(det.body_curr)
        if det.face_curr is not None:
            self.push_face(det.face_curr)

    def re_activate(self, det: "TrackRec", frame_id: int):     # D:570-584 (new_id=False)
        self._absorb(det)
        self.tracklet_len = 0
        self.state = TRACKED
        self.is_activated = True
        self.frame_id = frame_id
        self.score = det.score
        self.det_index = det.det_index

    def update(self, det: "TrackRec", frame_id: int):          # D:586-610
        self.frame_id = frame_id
        self.tracklet_len += 1
        self._absorb(det)
        self.state = TRACKED
        self.is_activated = True
        self.score = det.score
        self.det_index = det.det_index


def _joint(a: List[TrackRec], b: List[TrackRec]) -> List[TrackRec]:       # D:1642-1653
    seen = set()
    out = []
    for t in a:
        seen.add(t.track_id)
        out.append(t)
    for t in b:
        if t.track_id not in seen:
            seen.add(t.track_id)
            out.append(t)
    return out


def _sub(a: List[TrackRec], b: List[TrackRec]) -> List[TrackRec]:         # D:1655-1663
    keep = {}
    for t in a:
        keep[t.track_id] = t
    for t in b:
        # the reference tests `stracks.get(tid, 0)` -- an object is always truthy
        if t.track_id in keep:
            del keep[t.track_id]
    return list(keep.values())


class TrackerOracle:
    """Tracking half of BoTSORT.update (D:1414-1639) on pre-extracted arrays.

    One instance per video stream; the ID counter is per instance (the reference's counter is
    process-global and reset by every BoTSORT(), D:390/1264 -- one process per stream gives the
    same IDs).
    """

    def __init__(self, frame_rate: int = 30, faithful_iou: bool = False, motion_gate: bool = False):
        self.motion_gate = motion_gate      # NOT in the reference's loop (A6): see _gate_stage1
        self.tracked: List[TrackRec] = []
        self.lost: List[TrackRec] = []
        self.removed: List[TrackRec] = []
        self._count = 0
        self.frame_id = 0
        self.track_high_thresh = 0.40       # D:1268-1277
        self.track_low_thresh = 0.1
        self.new_track_thresh = 0.9
        self.match_thresh = 0.8
        self.track_buffer = 300
        self.feature_history = 300
        self.proximity_thresh = 0.5
        self.appearance_thresh = 0.25
        self.buffer_size = int(frame_rate / 30.0 * self.track_buffer)
        self.max_time_lost = self.buffer_size
        self.kf = KalmanOracle()
        self.faithful_iou = faithful_iou
        self.trace = None                   # filled per frame: dict of intermediate results

    def _next_id(self):
        self._count += 1
        return self._count

    def _iou(self, a: List[TrackRec], b: List[TrackRec]):
        return iou_cost([t.tlbr for t in a], [t.tlbr for t in b], faithful=self.faithful_iou)

    def _gate_stage1(self, dists, pool: List[TrackRec], dets: List[TrackRec]):
        """Optional extension (bts_tracker_config.motion_gate): upstream BoT-SORT's `gate_cost_matrix` on the stage-1
        costs, built from the reference's own KalmanFilter.gating_distance (D:338-380, 'maha', 4 dof) and
        chi2inv95[4] (D:139-149) -- the reference defines both and never calls them.  Gated entries get a cost
        no assignment can use (+inf upstream; a finite 1e9 here so that the dense JV's arithmetic stays finite)."""
        if not self.motion_gate or dists.size == 0:
            return dists
        meas = np.asarray([TrackRec.tlwh_to_xywh(t.tlwh0) for t in dets])
        dists = np.array(dists, dtype=np.float64)
        for i, t in enumerate(pool):
            d2 = self.kf.gating_distance(t.mean, t.cov, meas)
            dists[i, d2 > CHI2INV95[4]] = 1e9
        return dists

    def _multi_predict(self, pool: List[TrackRec]):          # D:524-536
        if not pool:
            return
        mean = np.asarray([t.mean.copy() for t in pool])
        cov = np.asarray([t.cov for t in pool])
        for i, t in enumerate(pool):
            if t.state != TRACKED:
                mean[i][6] = 0
                mean[i][7] = 0
        mean, cov = self.kf.multi_predict(mean, cov)
        for i, t in enumerate(pool):
            t.mean = mean[i]
            t.cov = cov[i]

    def update_arrays(self, boxes: np.ndarray, scores: np.ndarray, body_feats: np.ndarray,
                      face_feats: Optional[np.ndarray] = None) -> List[TrackRec]:
        """boxes[M,4] integer-pixel tlbr (what the detector emits, D:1009-1012), scores[M],
        body_feats[M,D] float32 L2-normalised, face_feats[M,Df] float32 or None.
        NOTE: like the reference (Q8) this renormalises rows of body_feats/face_feats in place."""
        self.frame_id += 1
        fid = self.frame_id
        activated: List[TrackRec] = []
        refind: List[TrackRec] = []
        lost_now: List[TrackRec] = []
        removed_now: List[TrackRec] = []
        M = len(boxes)
        boxes = np.asarray(boxes)
        tr = {}

        unconfirmed = [t for t in self.tracked if not t.is_activated]
        confirmed = [t for t in self.tracked if t.is_activated]
        pool = _joint(confirmed, self.lost)
        self._multi_predict(pool)
        N = len(pool)

        # similarities of every detection against the pool's *latest* features (Q1), computed
        # on the features as handed in (before the in-place renormalisation below)
        if M > 0 and N > 0:
            body_sim = cosine_similarity(np.asarray([t.body_curr for t in pool], dtype=np.float32), body_feats)
            if face_feats is not None:
                face_sim = face_similarity_fixup(cosine_similarity(
                    np.asarray([t.face_curr for t in pool], dtype=np.float32), face_feats))
            else:
                face_sim = np.zeros((N, M), dtype=np.float32)
        else:
            body_sim = np.zeros((N, M), dtype=np.float32)
            face_sim = np.zeros((N, M), dtype=np.float32)

        def make(i):
            b = boxes[i]
            tlwh = np.asarray([b[0], b[1], b[2] - b[0], b[3] - b[1]])
            return TrackRec(tlwh, scores[i], body_feats[i],
                            None if face_feats is None else face_feats[i], det_index=i)

        hi_idx = [i for i in range(M) if scores[i] > self.track_high_thresh]
        lo_idx = [i for i in range(M) if self.track_low_thresh <= scores[i] <= self.track_high_thresh]
        dets_hi = [make(i) for i in hi_idx]
        dets_lo = [make(i) for i in lo_idx]
        tr["hi_idx"] = np.asarray(hi_idx, dtype=np.int64)
        tr["lo_idx"] = np.asarray(lo_idx, dtype=np.int64)

        # (test support: the boxes every stage's IoU matrix is built from, as float64 holders + "was float32" flags -- what the
        #  device keeps in pool_tlbr / det_tlbr / r2_tlbr / unc_tlbr; tests/test_lap_exact_host.py feeds them to the product's
        #  box cost source)
        det_tlbr = np.zeros((M, 4), dtype=np.float64)
        for lst, idx in ((dets_hi, hi_idx), (dets_lo, lo_idx)):
            for t, i in zip(lst, idx):
                det_tlbr[i] = t.tlbr
        tr["det_tlbr"] = det_tlbr
        tr["pool_tlbr"], tr["pool_f32"] = _as_box_array([t.tlbr for t in pool])

        # ---- stage 1: pool x high-score detections, D:1539-1566
        iou_d = self._iou(pool, dets_hi)
        if N > 0 and M > 0 and len(dets_hi) == 0:
            # the reference raises here when M >= 2 (quirk Q19); defined as "no matches"
            dists = iou_d
        else:
            dists = fuse_cost_stage1(iou_d, body_sim[:, hi_idx], face_sim[:, hi_idx],
                                     self.proximity_thresh, self.appearance_thresh)
        dists = self._gate_stage1(dists, pool, dets_hi)
        m1, u_trk, u_det = linear_assignment(dists, self.match_thresh)
        tr["cost1"] = dists
        tr["iou1"] = iou_d
        tr["m1"] = np.asarray(m1, dtype=np.int64).reshape(-1, 2)
        tr["u_trk1"] = np.asarray(u_trk, dtype=np.int64)
        tr["u_det1"] = np.asarray(u_det, dtype=np.int64)
        for it, idet in tr["m1"]:
            t = pool[it]
            if t.state == TRACKED:
                t.update(dets_hi[idet], fid)
                activated.append(t)
            else:
                t.re_activate(dets_hi[idet], fid)
                refind.append(t)

        # ---- stage 2: still-tracked leftovers x low-score detections, D:1568-1586
        r_tracked = [pool[i] for i in u_trk if pool[i].state == TRACKED]
        tr["r2_tlbr"], tr["r2_f32"] = _as_box_array([t.tlbr for t in r_tracked])
        dists2 = self._iou(r_tracked, dets_lo)
        m2, u_trk2, u_det2 = linear_assignment(dists2, 0.5)
        tr["cost2"] = dists2
        tr["m2"] = np.asarray(m2, dtype=np.int64).reshape(-1, 2)
        tr["u_trk2"] = np.asarray(u_trk2, dtype=np.int64)
        tr["u_det2"] = np.asarray(u_det2, dtype=np.int64)
        for it, idet in tr["m2"]:
            t = r_tracked[it]
            if t.state == TRACKED:
                t.update(dets_lo[idet], fid)
                activated.append(t)
            else:
                t.re_activate(dets_lo[idet], fid)
                refind.append(t)
        for it in u_trk2:
            t = r_tracked[it]
            if t.state != LOST:
                t.state = LOST
                lost_now.append(t)

        # ---- stage 3: unconfirmed x leftover high-score detections, D:1588-1612
        left = [dets_hi[i] for i in u_det]
        tr["unc_tlbr"], tr["unc_f32"] = _as_box_array([t.tlbr for t in unconfirmed])
        tr["left_det"] = np.asarray([d.det_index for d in left], dtype=np.int64)
        iou3 = self._iou(unconfirmed, left)
        D = body_feats.shape[1] if body_feats.ndim == 2 else 0
        uf = (np.asarray([t.body_curr for t in unconfirmed], dtype=np.float32)
              if unconfirmed else np.zeros((0, D), dtype=np.float32))
        lf = (np.asarray([t.body_curr for t in left], dtype=np.float32)
              if left else np.zeros((0, D), dtype=np.float32))
        dists3 = fuse_cost_stage3(iou3, cosine_similarity(uf, lf),
                                  self.proximity_thresh, self.appearance_thresh)
        m3, u_unc, u_det3 = linear_assignment(dists3, 0.7)
        tr["cost3"] = dists3
        tr["iou3"] = iou3
        tr["m3"] = np.asarray(m3, dtype=np.int64).reshape(-1, 2)
        tr["u_trk3"] = np.asarray(u_unc, dtype=np.int64)
        tr["u_det3"] = np.asarray(u_det3, dtype=np.int64)
        for it, idet in tr["m3"]:
            unconfirmed[it].update(left[idet], fid)
            activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED
            removed_now.append(unconfirmed[it])

        # ---- births, D:1614-1621
        for i in u_det3:
            d = left[i]
            if d.score < self.new_track_thresh:
                continue
            d.activate(self.kf, fid, self)
            activated.append(d)

        # ---- expiry, D:1623-1627
        for t in self.lost:
            if fid - t.end_frame > self.max_time_lost:
                t.state = REMOVED
                removed_now.append(t)

        # ---- merge, D:1629-1637 (order matters: it defines next frame's row indices)
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        self.tracked = _joint(self.tracked, activated)
        self.tracked = _joint(self.tracked, refind)
        self.lost = _sub(self.lost, self.tracked)
        self.lost.extend(lost_now)
        self.lost = _sub(self.lost, self.removed)        # *previous* removals only (Q6)
        self.removed.extend(removed_now)
        self.tracked, self.lost = self._dedupe(self.tracked, self.lost)
        self.trace = tr
        return self.tracked

    def _dedupe(self, a: List[TrackRec], b: List[TrackRec]):    # D:1665-1680
        pd = self._iou(a, b)
        ps, qs = np.where(pd < 0.15)
        drop_a, drop_b = set(), set()
        for p, q in zip(ps, qs):
            tp = a[p].frame_id - a[p].start_frame
            tq = b[q].frame_id - b[q].start_frame
            if tp > tq:
                drop_b.add(int(q))
            else:
                drop_a.add(int(p))
        return ([t for i, t in enumerate(a) if i not in drop_a],
                [t for i, t in enumerate(b) if i not in drop_b])

    # ---- snapshot helpers used by tests / fixtures
    @staticmethod
    def table(tracks: List[TrackRec]) -> dict:
        return {
            "track_id": np.asarray([t.track_id for t in tracks], dtype=np.int64),
            "state": np.asarray([t.state for t in tracks], dtype=np.int64),
            "is_activated": np.asarray([t.is_activated for t in tracks], dtype=np.int64),
            "frame_id": np.asarray([t.frame_id for t in tracks], dtype=np.int64),
            "start_frame": np.asarray([t.start_frame for t in tracks], dtype=np.int64),
            "tracklet_len": np.asarray([t.tracklet_len for t in tracks], dtype=np.int64),
            "mean": (np.asarray([np.asarray(t.mean, dtype=np.float64) for t in tracks]).reshape(-1, 8)),
        }
