"""Association functions with the reference's names and return conventions
(demo_bottrack_onnx_tflite.py:1682-1761), computed by HIP kernels through the C ABI
(csrc/assoc.hip, csrc/cosine.hip, csrc/lap.hip).  Inputs may be lists of STrack-like objects
(anything with `.tlbr`) or lists / arrays of tlbr boxes, as in the reference (D:1753-1758).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import _lib


def _dev(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _boxes(items) -> "tuple[np.ndarray, np.ndarray]":
    """-> float64 [K,4] boxes and uint8 [K] 'was float32' flags (the kernels reproduce NumPy's
    scalar dtype flow from them, include/botsort_hip.h conventions)."""
    k = len(items)
    out = np.empty((k, 4), dtype=np.float64)
    f32 = np.zeros(k, dtype=np.uint8)
    for i, it in enumerate(items):
        b = np.asarray(it.tlbr if hasattr(it, "tlbr") else it)
        out[i] = b
        f32[i] = 1 if b.dtype == np.float32 else 0
    return out, f32


def bbox_ious(atlbrs, btlbrs) -> np.ndarray:
    """D:1731-1743 -- IoU matrix (not the cost)."""
    return 1 - _iou_cost(atlbrs, btlbrs)


def _iou_cost(a_items, b_items) -> np.ndarray:
    import torch
    n, m = len(a_items), len(b_items)
    if n == 0 or m == 0:
        return 1 - np.zeros((n, m), dtype=np.float32)            # D:1739-1741
    lib = _lib.load(require_gpu=True)
    a, af = _boxes(a_items)
    b, bf = _boxes(b_items)
    da, daf, db, dbf = _dev(a, np.float64), _dev(af, np.uint8), _dev(b, np.float64), _dev(bf, np.uint8)
    cost = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _lib.check(lib.bts_iou_distance(_lib.default_context().handle, _lib.ptr(da), _lib.ptr(daf), n, _lib.ptr(db),
                                    _lib.ptr(dbf), m, _lib.ptr(cost), m, _lib.current_stream()))
    return cost.cpu().numpy()


def iou_distance(atracks, btracks) -> np.ndarray:
    """D:1745-1761: cost = 1 - IoU."""
    return _iou_cost(atracks, btracks)


def bbox_iou(atlbr, btlbr) -> float:
    """D:1695-1713 (one pair; goes through the same kernel)."""
    return float(1 - _iou_cost([np.asarray(atlbr)], [np.asarray(btlbr)])[0, 0])


def cosine_similarity(track_feats, det_feats) -> np.ndarray:
    """max(0, T @ D.T) in float32 -- the D:1599 definition."""
    import torch
    t = np.ascontiguousarray(track_feats, dtype=np.float32)
    d = np.ascontiguousarray(det_feats, dtype=np.float32)
    n, m = len(t), len(d)
    if n == 0 or m == 0:
        return np.zeros((n, m), dtype=np.float32)
    lib = _lib.load(require_gpu=True)
    dt, dd = _dev(t, np.float32), _dev(d, np.float32)
    sim = torch.empty((n, m), dtype=torch.float32, device="cuda")
    _lib.check(lib.bts_cos_similarity(_lib.default_context().handle, _lib.ptr(dt), n, _lib.ptr(dd), m, t.shape[1],
                                      _lib.ptr(sim), m, _lib.current_stream()))
    return sim.cpu().numpy()


def fuse_cost_stage1(iou_d, body_sim, face_sim=None, proximity_thresh=0.5, appearance_thresh=0.25) -> np.ndarray:
    """D:1539-1554."""
    import torch
    n, m = iou_d.shape
    if n == 0 or m == 0:
        return np.zeros((n, m))
    lib = _lib.load(require_gpu=True)
    di, db = _dev(iou_d, np.float64), _dev(body_sim, np.float32)
    df = None if face_sim is None else _dev(face_sim, np.float32)
    out = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _lib.check(lib.bts_fuse_cost_stage1(_lib.default_context().handle, _lib.ptr(di), m, _lib.ptr(db), _lib.ptr(df), m,
                                        n, m, proximity_thresh, appearance_thresh, _lib.ptr(out), m,
                                        _lib.current_stream()))
    return out.cpu().numpy()


def fuse_cost_stage3(iou_d, body_sim, proximity_thresh=0.5, appearance_thresh=0.25) -> np.ndarray:
    """D:1590-1602."""
    import torch
    n, m = iou_d.shape
    if n == 0 or m == 0:
        return np.zeros((n, m))
    lib = _lib.load(require_gpu=True)
    di, db = _dev(iou_d, np.float64), _dev(body_sim, np.float32)
    out = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _lib.check(lib.bts_fuse_cost_stage3(_lib.default_context().handle, _lib.ptr(di), m, _lib.ptr(db), m, n, m,
                                        proximity_thresh, appearance_thresh, _lib.ptr(out), m,
                                        _lib.current_stream()))
    return out.cpu().numpy()


# The reference's behaviour on EXACT cost ties (lap.lapjv's dense JV scan orders) through bts_lapjv_exact: certificate +
# replay (csrc/lap_exact.h).  Off by default until the device path has been run on a GPU (round 3: access closed); the fast
# solver is index-identical whenever the optimum is unique.
EXACT_TIES = False
last_tie_info = None       # (certificate said tied, replay ran, error flags) of the last lapjv(exact_ties=True)


def lapjv(cost, extend_cost=True, cost_limit=np.inf, exact_ties=None, force_replay=False):
    """`lap.lapjv(cost, extend_cost=True, cost_limit=L)` as the reference calls it (D:1686):
    returns (opt, x, y).  `exact_ties` (default: module switch EXACT_TIES): also reproduce the reference on exact ties."""
    global last_tie_info
    import torch
    if not extend_cost or not np.isfinite(cost_limit):
        raise NotImplementedError("only lapjv(extend_cost=True, cost_limit=<finite>) is provided")
    c = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = c.shape
    lib = _lib.load(require_gpu=True)
    dc = _dev(c, np.float64)
    x = torch.empty((max(n, 1),), dtype=torch.int32, device="cuda")
    y = torch.empty((max(m, 1),), dtype=torch.int32, device="cuda")
    if EXACT_TIES if exact_ties is None else exact_ties:
        info = np.zeros(3, dtype=np.int32)
        _lib.check(lib.bts_lapjv_exact(_lib.default_context().handle, _lib.ptr(dc), n, m, m, float(cost_limit), _lib.ptr(x),
                                       _lib.ptr(y), 1 if force_replay else 0, info.ctypes.data, _lib.current_stream()))
        last_tie_info = tuple(int(v) for v in info)
        if info[2]:
            raise _lib.BtsError("bts_lapjv_exact: the exact-tie step reported error flags 0x%x" % int(info[2]))
    else:
        _lib.check(lib.bts_lapjv(_lib.default_context().handle, _lib.ptr(dc), n, m, m, float(cost_limit), _lib.ptr(x),
                                 _lib.ptr(y), _lib.current_stream()))
    xs, ys = x[:n].cpu().numpy(), y[:m].cpu().numpy()
    rows = np.nonzero(xs >= 0)[0]
    opt = float(c[rows, xs[rows]].sum())
    return opt, xs, ys


def linear_assignment(cost_matrix: np.ndarray, thresh: float):
    """D:1682-1693, including its return-type quirks (Q14)."""
    if cost_matrix.size == 0:
        return (np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])),
                tuple(range(cost_matrix.shape[1])))
    _, x, y = lapjv(cost_matrix, extend_cost=True, cost_limit=thresh)
    matches = np.asarray([[ix, mx] for ix, mx in enumerate(x) if mx >= 0])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]


def feature_ema(smooth, feature: np.ndarray, alpha: float = 0.9):
    """One STrack.update_body_features / update_face_features step (D:492-514) through the C-ABI kernels ->
    (smooth', feature).  First call (smooth is None): the reference aliases smooth = feature and normalises IN PLACE
    (quirk Q8) -- `feature` (a writable float32 array) is overwritten with its renormalised self and returned as both.
    Later calls: smooth' = normalise(alpha * smooth + (1 - alpha) * feature), a new array; feature untouched."""
    import torch
    lib = _lib.load(require_gpu=True)
    ctx = _lib.default_context().handle
    d = int(feature.shape[-1])
    f = _dev(feature.reshape(1, d), np.float32)
    if smooth is None:
        _lib.check(lib.bts_feature_renorm(ctx, _lib.ptr(f), None, 1, d, _lib.current_stream()))
        feature[...] = f.cpu().numpy().reshape(feature.shape)
        return feature, feature
    s = _dev(np.asarray(smooth, dtype=np.float32).reshape(1, d), np.float32)
    curr = torch.empty_like(f)
    has = torch.ones((1,), dtype=torch.uint8, device="cuda")
    _lib.check(lib.bts_feature_ema(ctx, _lib.ptr(curr), _lib.ptr(s), _lib.ptr(has), None, _lib.ptr(f), None, 1, d,
                                   float(alpha), _lib.current_stream()))
    return s.cpu().numpy().reshape(feature.shape), feature


def dedupe_flags(a_items, b_items, age_a: Sequence[int], age_b: Sequence[int], thresh: float = 0.15):
    """The pair test of remove_duplicate_stracks (D:1665-1680) -> (drop_a[n], drop_b[m]) bool."""
    import torch
    n, m = len(a_items), len(b_items)
    if n == 0 or m == 0:
        return np.zeros(n, dtype=bool), np.zeros(m, dtype=bool)
    lib = _lib.load(require_gpu=True)
    a, af = _boxes(a_items)
    b, bf = _boxes(b_items)
    da, daf, db, dbf = _dev(a, np.float64), _dev(af, np.uint8), _dev(b, np.float64), _dev(bf, np.uint8)
    ga, gb = _dev(age_a, np.int32), _dev(age_b, np.int32)
    fa = torch.empty((n,), dtype=torch.uint8, device="cuda")
    fb = torch.empty((m,), dtype=torch.uint8, device="cuda")
    _lib.check(lib.bts_dedupe_pairs(_lib.default_context().handle, _lib.ptr(da), _lib.ptr(daf), _lib.ptr(ga), n,
                                    _lib.ptr(db), _lib.ptr(dbf), _lib.ptr(gb), m, float(thresh), _lib.ptr(fa),
                                    _lib.ptr(fb), _lib.current_stream()))
    return fa.cpu().numpy().astype(bool), fb.cpu().numpy().astype(bool)


def associate_parts(base_boxes, base_centres, target_boxes, target_centres, used=None, picks: int = 1):
    """The greedy part attachment of D:1379-1411 for a whole frame: find_most_relevant_object
    (D:1763-1791) for every base box in order, `picks` picks each.  Integer (x1, y1, x2, y2) boxes and
    (cx, cy) centres; `used` marks targets already taken.  -> (idx[nb, picks] with -1 = none, used[nt])."""
    import torch
    bb = np.ascontiguousarray(np.asarray(base_boxes, dtype=np.int32).reshape(-1, 4))
    tb = np.ascontiguousarray(np.asarray(target_boxes, dtype=np.int32).reshape(-1, 4))
    nb, nt = len(bb), len(tb)
    u = np.zeros(nt, dtype=np.uint8) if used is None else np.asarray(used).astype(np.uint8)
    if nb == 0:
        return np.zeros((0, picks), dtype=np.int64), u.astype(bool)
    lib = _lib.load(require_gpu=True)
    dbb, dbc = _dev(bb, np.int32), _dev(np.asarray(base_centres).reshape(-1, 2), np.int32)
    dtb, dtc = _dev(tb, np.int32), _dev(np.asarray(target_centres).reshape(-1, 2), np.int32)
    du = _dev(u, np.uint8)
    out = torch.empty((nb * picks,), dtype=torch.int32, device="cuda")
    _lib.check(lib.bts_associate_parts(_lib.default_context().handle, _lib.ptr(dbb), _lib.ptr(dbc), nb, _lib.ptr(dtb),
                                       _lib.ptr(dtc), nt, _lib.ptr(du), int(picks), _lib.ptr(out), _lib.current_stream()))
    return out.cpu().numpy().astype(np.int64).reshape(nb, picks), du.cpu().numpy().astype(bool)


def bbox_iou_by_box(base_obj, target_obj) -> float:
    """D:1715-1729: IoU of two detector boxes (objects with integer x1, y1, x2, y2); 0.0 when they do not overlap.
    Goes through the IoU kernel on float64 copies of the pixel coordinates (exact there)."""
    box = lambda o: np.asarray([o.x1, o.y1, o.x2, o.y2], dtype=np.float64)
    return bbox_iou(box(base_obj), box(target_obj))


def find_most_relevant_object(base_obj, target_objs):
    """D:1763-1791 with the reference's signature: the not-yet-used target (objects with x1, y1, x2, y2,
    cx, cy, is_used) that overlaps `base_obj` most; it is marked used and returned (None if none overlaps)."""
    if not target_objs:
        return None
    box = lambda o: [o.x1, o.y1, o.x2, o.y2]
    idx, _ = associate_parts([box(base_obj)], [[base_obj.cx, base_obj.cy]], [box(t) for t in target_objs],
                             [[t.cx, t.cy] for t in target_objs], used=[bool(t.is_used) for t in target_objs])
    if idx[0, 0] < 0:
        return None
    hit = target_objs[int(idx[0, 0])]
    hit.is_used = True
    return hit
