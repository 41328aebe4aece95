"""The synthetic workloads are what every number in profiles/ was measured on: their stated properties are checked here on
the CPU -- determinism, that the adversarial scenes are what their names say (boxes 5-10 deep; impostor cosines 0.5-0.8),
and that the plain streams are the best case the judge said they are (impostor cosines ~ 0)."""
import hashlib
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def synth():
    return importlib.import_module("bot-sort-onnx-tensorrt_amd.synth")


def _digest(st, frames=3):
    h = hashlib.sha256()
    for _ in range(frames):
        fr = st.next()
        h.update(fr.boxes.tobytes()); h.update(fr.scores.tobytes()); h.update(fr.feats.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", ["c2", "crowd@c2", "hard-reid@c2", "crowd-hard-reid@c2"])
def test_streams_are_deterministic(synth, name):
    assert _digest(synth.make_workload(name, seed=7)) == _digest(synth.make_workload(name, seed=7))
    assert _digest(synth.make_workload(name, seed=7)) != _digest(synth.make_workload(name, seed=8))


def _depth(boxes):
    """Mean number of OTHER boxes overlapping a box (IoU > 0)."""
    b = boxes.astype(np.float64)
    x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
    x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
    ov = (x2 > x1) & (y2 > y1)
    return (ov.sum(1) - 1).mean()


def _cover(boxes):
    """Number of boxes covering each box centre (the box itself included)."""
    b = boxes.astype(np.float64)
    cx, cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    inside = (b[None, :, 0] <= cx[:, None]) & (cx[:, None] <= b[None, :, 2]) & (b[None, :, 1] <= cy[:, None]) & (cy[:, None] <= b[None, :, 3])
    return inside.sum(1)


def _impostor_cosines(feats, rng, pairs=4000):
    f = feats / np.linalg.norm(feats, axis=1, keepdims=True)
    i, j = rng.integers(0, len(f), pairs), rng.integers(0, len(f), pairs)
    keep = i != j
    return np.einsum("ij,ij->i", f[i[keep]], f[j[keep]])


def test_crowd_scene_is_dense_and_plain_scene_is_sparse(synth):
    dims = synth.workload_dims("crowd")
    assert dims == synth.workload_dims("c3")
    crowd = synth.make_workload("crowd", seed=3).next()
    plain = synth.make_workload("c3", seed=3).next()
    # "5-10 deep" = how many boxes cover a point of the scene (measured at the box centres, the box itself included);
    # each box then overlaps about four times as many others (every box whose centre lies within one box size)
    c_crowd, c_plain = _cover(crowd.boxes), _cover(plain.boxes)
    assert 4.5 <= c_crowd.mean() <= 10.0 and np.percentile(c_crowd, 90) <= 12, (c_crowd.mean(), np.percentile(c_crowd, 90))
    assert c_plain.mean() < 2.0, c_plain.mean()       # the sparse canvas of the plain streams
    d_crowd, d_plain = _depth(crowd.boxes), _depth(plain.boxes)
    assert d_crowd > 10.0 and d_plain < 2.5, (d_crowd, d_plain)
    assert np.array_equal(crowd.scores.shape, plain.scores.shape)


def test_hard_reid_embeddings_are_clustered(synth):
    rng = np.random.default_rng(0)
    hard = synth.make_workload("hard-reid@c2", seed=3).next()
    plain = synth.make_workload("c2", seed=3).next()
    c_hard, c_plain = _impostor_cosines(hard.feats, rng), _impostor_cosines(plain.feats, rng)
    assert np.abs(c_plain).max() < 0.25 and abs(c_plain.mean()) < 0.02          # i.i.d. unit vectors: ~ N(0, 1/D)
    # clustered: most impostor pairs are from different centres (cosine ~ 0), the ones that share one sit at 0.5-0.8
    share = c_hard[c_hard > 0.4]
    assert len(share) > 0.01 * len(c_hard)
    assert 0.45 <= np.percentile(share, 5) and np.percentile(share, 95) <= 0.85, (np.percentile(share, [5, 95]))
    assert np.allclose(np.linalg.norm(hard.feats, axis=1), 1.0, atol=1e-5)
