"""The exact-tie code of the product (csrc/lap_exact.h: uniqueness certificate + phase-parallel replay of lap 0.4.0's dense
JV) compiled for the CPU by tests/host_emul with an executor that runs a workgroup's threads one after the other, against
the oracle's dense JV (oracle/lapjv_oracle.c) on tie families: INDEX vectors, not objectives.

  * replay (forced): x, y identical to the dense JV's for every case, every emulated workgroup size, both thread orders, any
    order of the edge list;
  * certificate: whenever it says "unique", the assignment it was shown IS the dense JV's; it says "tied" on the tie families
    that fool the fast solver (so the replay runs there) and "unique" on tie-free float costs (so it does not run there).
The same source runs on the device as k_lap_exact (tests/test_zz_gpu_ties.py compares THAT with the oracle on the GPU)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import bts_oracle as bo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lapx():
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s"])
    lib = ctypes.CDLL(os.path.join(d, "liblapx_host.so"))
    P = ctypes.c_void_p
    lib.lapx_host_exact.argtypes = [ctypes.c_int, ctypes.c_int, P, ctypes.c_int, ctypes.c_double, P, P, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, P, ctypes.c_uint, P]

    def run(cost, L, x, y, nth=64, order=0, force=False, base=None, shuffle=0):
        c = np.ascontiguousarray(cost, dtype=np.float64)
        n, m = c.shape
        x = np.ascontiguousarray(x, dtype=np.int32).copy()
        y = np.ascontiguousarray(y, dtype=np.int32).copy()
        info = np.zeros(4, dtype=np.int32)
        b = None if base is None else np.ascontiguousarray(base, dtype=np.float64)
        rc = lib.lapx_host_exact(n, m, c.ctypes.data, m, L, x.ctypes.data, y.ctypes.data, nth, order, int(force),
                                 None if b is None else b.ctypes.data, shuffle, info.ctypes.data)
        assert rc == 0 and info[1] == 0, (rc, info)
        return x, y, {"tied": bool(info[0]), "replayed": bool(info[2]), "edges": int(info[3])}
    return run


def scipy_visible(c, L):
    """SOME optimum of the extended problem (not the JV's on ties): what a fast exact solver would hand over."""
    n, m = c.shape
    ext = np.full((n + m, n + m), L / 2.0)
    ext[n:, m:] = 0
    ext[:n, :m] = c
    r, q = linear_sum_assignment(ext)
    x, y = np.full(n, -1, np.int32), np.full(m, -1, np.int32)
    for i, j in zip(r, q):
        if i < n and j < m:
            x[i], y[j] = j, i
    return x, y


def jv(c, L):
    _, x, y = bo.lapjv(np.ascontiguousarray(c, dtype=np.float64), extend_cost=True, cost_limit=L)
    return np.asarray(x, dtype=np.int32), np.asarray(y, dtype=np.int32)


def tie_families(rng, count, max_dim=10):
    L = 0.8
    for _ in range(count):
        kind = int(rng.integers(0, 8))
        n, m = (int(v) for v in rng.integers(1, max_dim, 2))
        if kind == 0:
            c = rng.integers(1, 6, (n, m)) / 5.0                       # integer grid: many ties, some at the limit
        elif kind == 1:
            c = rng.uniform(0.05, 0.95, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0      # tie-free floats
        elif kind == 2:
            c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
            j = rng.integers(0, m); c = np.concatenate([c, c[:, j:j + 1]], 1)            # duplicated detection column
        elif kind == 3:
            c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
            i = rng.integers(0, n); c = np.concatenate([c, c[i:i + 1]], 0)               # duplicated track row
        elif kind == 4:
            c = np.full((n, m), rng.choice([0.25, 0.4, 0.8])); c[rng.random((n, m)) < 0.3] = 1.0     # all-equal blocks, also at L/2 and L
        elif kind == 5:
            c = rng.choice([0.1, 0.3, 0.4, 0.5, 0.7, 0.8, 1.0], (n, m))
        elif kind == 6:
            c = rng.choice([0.2, 0.6], (n, m)).astype(np.float64); c[rng.random((n, m)) < 0.6] = rng.choice([0.85, 0.9, 1.0])   # non-edges of several values
        else:
            c = np.round(rng.uniform(0.0, 1.0, (n, m)), 1)                               # one decimal: ties everywhere, zeros included
        yield np.ascontiguousarray(c, dtype=np.float64), L


def test_forced_replay_equals_the_dense_jv_on_tie_families(lapx):
    rng = np.random.default_rng(0)
    for k, (c, L) in enumerate(tie_families(rng, 2500)):
        want = jv(c, L)
        x0, y0 = scipy_visible(c, L)
        for nth, order in ((64, 0), (7, 1), (1, 0), (256, 1)) if k % 5 == 0 else ((32, k & 1),):
            x, y, info = lapx(c, L, x0, y0, nth=nth, order=order, force=True, shuffle=k % 7)
            assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (k, nth, order, c, x, want[0])


def test_forced_replay_on_larger_sparse_problems(lapx):
    """Tracker-shaped problems (few feasible edges, many rows without any, non-edges at several costs > L), with planted ties:
    the regime in which lap's column list is rotated thousands of times per search (apply_runs)."""
    rng = np.random.default_rng(1)
    for k in range(40):
        n, m = int(rng.integers(20, 140)), int(rng.integers(8, 60))
        c = np.ones((n, m))
        rows = rng.permutation(n)[:min(n, m)]
        for j, i in enumerate(rows):
            c[i, j % m] = np.round(rng.uniform(0.05, 0.6), 2 if k % 2 else 6)
        extra = int(1.5 * m)
        c[rng.integers(0, n, extra), rng.integers(0, m, extra)] = np.round(rng.uniform(0.1, 0.79, extra), 1 if k % 3 == 0 else 6)
        far = (c == 1.0) & (rng.random((n, m)) < 0.05)
        c[far] = rng.uniform(0.81, 0.99, int(far.sum()))
        if k % 4 == 0:
            c[:, -1] = c[:, 0]
        want = jv(c, 0.8)
        x0, y0 = scipy_visible(c, 0.8)
        x, y, _ = lapx(c, 0.8, x0, y0, nth=64 if k % 2 else 256, order=k & 1, force=True, shuffle=k)
        assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), k


def test_certificate_is_sound_and_sharp(lapx):
    rng = np.random.default_rng(2)
    unique = tied = tied_and_scipy_differs = 0
    for k, (c, L) in enumerate(tie_families(rng, 4000)):
        want = jv(c, L)
        x0, y0 = scipy_visible(c, L)
        x, y, info = lapx(c, L, x0, y0, nth=64, order=k & 1, shuffle=k % 5)
        # the end result is ALWAYS the dense JV's: either certified as is, or replayed
        assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (k, info, c)
        assert info["replayed"] == info["tied"]
        if info["tied"]:
            tied += 1
            tied_and_scipy_differs += not (np.array_equal(x0, want[0]) and np.array_equal(y0, want[1]))
        else:
            unique += 1
            assert np.array_equal(x0, want[0]) and np.array_equal(y0, want[1]), (k, c)      # soundness
        # the verdict does not depend on which optimum it is shown
        _, _, info2 = lapx(c, L, want[0], want[1], nth=32, order=0)
        assert info2["tied"] == info["tied"], (k, c)
    assert unique > 800 and tied > 800 and tied_and_scipy_differs > 300


def test_certificate_does_not_fire_on_tie_free_costs(lapx):
    rng = np.random.default_rng(3)
    for k in range(150):
        n, m = (int(v) for v in rng.integers(5, 80, 2))
        c = rng.uniform(0.05, 0.95, (n, m)); c[rng.random((n, m)) < 0.8] = 1.0
        x0, y0 = scipy_visible(c, 0.8)
        x, y, info = lapx(c, 0.8, x0, y0, nth=64, shuffle=k)
        assert not info["tied"] and np.array_equal(x, x0) and np.array_equal(y, y0), k
    # cost exactly at the limit: matched or not costs the same -> tied
    _, _, info = lapx(np.full((1, 1), 0.8), 0.8, [0], [0])
    assert info["tied"]
    _, _, info = lapx(np.full((1, 1), 0.8), 0.8, [-1], [-1])
    assert info["tied"]
    _, _, info = lapx(np.full((1, 1), 0.3), 0.8, [0], [0])
    assert not info["tied"]


def test_patched_source_equals_the_dense_source(lapx):
    """The tracker's cost source: non-edges come from a BASE matrix (the IoU cost), feasible edges from the edge list (where
    the fused appearance term may have lowered them).  Replaying through it must give what the full matrix gives."""
    rng = np.random.default_rng(4)
    for k in range(300):
        n, m = (int(v) for v in rng.integers(1, 12, 2))
        base = np.round(rng.uniform(0.3, 1.0, (n, m)), 1)               # "IoU costs": ties, some <= L
        fused = base.copy()
        hit = rng.random((n, m)) < 0.25
        fused[hit] = np.minimum(fused[hit], np.round(rng.uniform(0.0, 0.25, int(hit.sum())), 2))    # appearance lowers some entries
        want = jv(fused, 0.8)
        x0, y0 = scipy_visible(fused, 0.8)
        x, y, _ = lapx(fused, 0.8, x0, y0, nth=16, order=k & 1, force=True, base=base, shuffle=k)
        assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), k


def test_empty_and_degenerate_shapes(lapx):
    for n, m in ((1, 1), (1, 5), (5, 1), (3, 3)):
        c = np.ones((n, m))                                              # no feasible edge at all
        x, y, info = lapx(c, 0.8, np.full(n, -1), np.full(m, -1), force=True)
        assert (x == -1).all() and (y == -1).all() and not info["tied"]
    c = np.zeros((4, 4))                                                 # everything free and equal
    want = jv(c, 0.8)
    x, y, info = lapx(c, 0.8, *scipy_visible(c, 0.8))
    assert info["tied"] and np.array_equal(x, want[0]) and np.array_equal(y, want[1])


def test_box_cost_source_premise_on_tracker_streams(lapx):
    """What tie_mode 1 assumes inside the tracker (k_lap_exact's BoxSrc): an entry of a stage's cost matrix that is NOT a feasible
    edge (cost > the stage's match threshold) is exactly the pair's IoU cost -- the appearance term only ever lowers entries to
    <= appearance_thresh, which makes them edges (D:1539-1554, D:1590-1602).  Checked on the oracle's own stage matrices of
    plain / churn / dense-crowd / dup-box streams, and the replay through (IoU base + feasible edges) must equal the dense JV on
    the full fused matrix.  (With face features the premise can fail -- a pair whose FACE passes the gate keeps its BODY distance
    even above the threshold -- which is why bts_tracker_create refuses tie_mode 1 with face_dim > 0.)"""
    import importlib
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    checked = tied = overrides = 0
    for kw in (dict(churn=0.1), dict(side_scale=40.0), dict(dup_frac=0.1, churn=0.05), dict(dup_frac=0.05, side_scale=60.0),
               dict(dup_frac=0.1, churn=0.05, face_dim=32), dict(face_dim=32, side_scale=60.0)):
        st = pkg.synth.SyntheticStream(96, 48, 64, seed=31, **kw)
        ora = bo.TrackerOracle()
        for f in range(10):
            fr = st.next()
            ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(),
                              None if fr.face_feats is None else fr.face_feats.copy())
            for k, L, base_key in ((1, 0.8, "iou1"), (2, 0.5, "cost2"), (3, 0.7, "iou3")):
                c, base = ora.trace["cost%d" % k], ora.trace[base_key]
                if c.size == 0:
                    continue
                c = np.ascontiguousarray(c, dtype=np.float64); base = np.ascontiguousarray(base, dtype=np.float64)
                far = c > L
                if not kw.get("face_dim"):
                    assert np.array_equal(c[far], base[far]), (kw, f, k)             # the premise (body features only)
                else:
                    # with faces a far entry may be a body distance below the IoU cost: those travel as overrides (stage 1 only)
                    overrides += int((c[far] != base[far]).sum())
                    assert k == 1 or np.array_equal(c[far], base[far]), (kw, f, k)
                assert (c <= base).all()                                              # the appearance term only ever lowers an entry
                want = jv(c, L)
                x0, y0 = scipy_visible(c, L)
                x, y, info = lapx(c, L, x0, y0, nth=64, order=f & 1, base=base, shuffle=f)
                assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (kw, f, k)
                assert np.array_equal(x, ora.trace["x%d" % k]) if ("x%d" % k) in ora.trace else True
                checked += 1; tied += info["tied"]
    assert checked > 120 and tied > 15 and overrides > 5, (checked, tied, overrides)


def test_near_ties_around_the_tolerance(lapx):
    """Ties broken by tiny perturbations, from far below the certificate's tolerance (1e-9) to far above it: whatever the
    verdict, the end result is the dense JV's -- below the tolerance the replay runs (and, being the same arithmetic, follows
    the JV through its own rounding), above it the optimum is unique for every exact solver."""
    rng = np.random.default_rng(5)
    verdicts = {}
    for eps in (0.0, 1e-15, 1e-12, 1e-10, 5e-10, 3e-9, 1e-8, 1e-7, 1e-5):
        tied = 0
        for k in range(150):
            n, m = (int(v) for v in rng.integers(2, 9, 2))
            c = rng.integers(1, 5, (n, m)) / 5.0
            c = c + eps * rng.integers(-3, 4, (n, m))              # exact multiples of eps: ties of the grid survive or split
            c = np.clip(c, 0.0, 1.0)
            want = jv(c, 0.8)
            x, y, info = lapx(c, 0.8, *scipy_visible(c, 0.8), nth=32, order=k & 1, shuffle=k)
            assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (eps, k, c)
            tied += info["tied"]
        verdicts[eps] = tied
    assert verdicts[0.0] > 80 and verdicts[1e-12] > 80            # below the tolerance: still "tied"
    assert verdicts[1e-5] < verdicts[0.0]                          # far above: mostly split


def test_replay_on_a_mid_size_problem_with_long_list_runs(lapx):
    """600 x 200 with duplicated columns and rows: thousands of list insertions per search, runs longer than the queue they
    rotate (apply_runs with r > q), several emulated workgroup sizes."""
    rng = np.random.default_rng(6)
    n, m = 600, 200
    c = np.ones((n, m))
    rows = rng.permutation(n)[:m]
    for j, i in enumerate(rows):
        c[i, j] = np.round(rng.uniform(0.05, 0.6), 2)
    extra = 2 * m
    c[rng.integers(0, n, extra), rng.integers(0, m, extra)] = np.round(rng.uniform(0.1, 0.79, extra), 1)
    far = (c == 1.0) & (rng.random((n, m)) < 0.02)
    c[far] = rng.uniform(0.81, 0.99, int(far.sum()))
    c[:, -3:] = c[:, :3]
    c[-2:] = c[[int(rows[4]), int(rows[9])]]
    want = jv(c, 0.8)
    x0, y0 = scipy_visible(c, 0.8)
    for nth, order in ((1024, 0), (100, 1)):
        x, y, info = lapx(c, 0.8, x0, y0, nth=nth, order=order, shuffle=3)
        assert info["tied"] and np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (nth, order)


def test_exact_tie_source_under_sanitizers():
    """The same source under AddressSanitizer + UBSan in a child interpreter (gcc's runtime preloaded): the harness leaves a
    poisoned gap behind every scratch array (and a sentinel pattern checked after each call), so an index running off the end
    of ONE of the 29 scratch arrays -- which on the device would silently land in its neighbour -- is an error here."""
    import sys
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "asan"])
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no gcc ASan runtime")
    code = r"""
import ctypes, numpy as np
from tests import test_lap_exact_host as t
lib = ctypes.CDLL(%r)
P = ctypes.c_void_p
lib.lapx_host_exact.argtypes = [ctypes.c_int, ctypes.c_int, P, ctypes.c_int, ctypes.c_double, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_uint, P]
def run(c, L, x, y, nth, order, force, shuffle):
    c = np.ascontiguousarray(c, dtype=np.float64); n, m = c.shape
    x = np.ascontiguousarray(x, dtype=np.int32).copy(); y = np.ascontiguousarray(y, dtype=np.int32).copy()
    info = np.zeros(4, dtype=np.int32)
    rc = lib.lapx_host_exact(n, m, c.ctypes.data, m, L, x.ctypes.data, y.ctypes.data, nth, order, int(force), None, shuffle, info.ctypes.data)
    assert rc == 0 and info[1] == 0, (rc, info)
    return x, y
rng = np.random.default_rng(11)
for k, (c, L) in enumerate(t.tie_families(rng, 700, max_dim=16)):
    want = t.jv(c, L)
    x, y = run(c, L, *t.scipy_visible(c, L), int(rng.choice([1, 3, 64, 1024])), k & 1, k %% 2 == 0, k %% 11)
    assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), k
for n, m in ((1, 1), (1, 7), (7, 1)):
    for v in (0.1, 0.8, 1.0):
        c = np.full((n, m), v)
        want = t.jv(c, 0.8)
        x, y = run(c, 0.8, *t.scipy_visible(c, 0.8), 64, 0, True, 1)
        assert np.array_equal(x, want[0]) and np.array_equal(y, want[1])
print("ok")
""" % os.path.join(d, "liblapx_host_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONPATH=os.path.dirname(HERE),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ok" in p.stdout, "rc=%d\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-6000:]


def test_overrides_above_the_limit(lapx):
    """Face features: a pair whose FACE passes the appearance gate keeps its BODY distance in the cost matrix even when that
    is above the match threshold (D:1543-1554) -- a non-edge that is not the IoU cost.  The device hands such entries to the
    exact-tie step as OVERRIDES (top end of the edge arrays); the replay through (IoU base + edges + overrides) must equal the
    dense JV on the full matrix, and the certificate must ignore them (nothing above the limit is a feasible edge)."""
    rng = np.random.default_rng(8)
    n_over = tied = 0
    for k in range(400):
        n, m = (int(v) for v in rng.integers(1, 12, 2))
        base = np.round(rng.uniform(0.3, 1.0, (n, m)), 1)               # "IoU costs" with ties
        fused = base.copy()
        hit = rng.random((n, m)) < 0.2
        fused[hit] = np.minimum(fused[hit], np.round(rng.uniform(0.0, 0.25, int(hit.sum())), 2))       # appearance edges
        face = (rng.random((n, m)) < 0.15) & (base > 0.85)
        fused[face] = np.round(rng.uniform(0.81, 0.84, int(face.sum())), 2)                              # body distance kept, above the limit
        n_over += int(((fused > 0.8) & (fused != base)).sum())
        want = jv(fused, 0.8)
        x0, y0 = scipy_visible(fused, 0.8)
        x, y, info = lapx(fused, 0.8, x0, y0, nth=16, order=k & 1, force=k % 2 == 0, base=base, shuffle=k)
        assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (k, info)
        assert (info["edges"] >> 16) == int(((fused > 0.8) & (fused != base)).sum())
        # the verdict is the one the full matrix gives without any notion of overrides
        _, _, info2 = lapx(fused, 0.8, x0, y0, nth=16)
        assert info2["tied"] == info["tied"], k
        tied += info["tied"]
    assert n_over > 300 and tied > 50


def test_tracker_box_source_on_the_cpu():
    """The code k_lap_exact runs inside the tracker -- csrc/lap_src.h BoxSrc over csrc/iou_math.h's iou::cost -- compiled by hipcc
    for the HOST and fed with the oracle's own stage data (pool / stage-2 / unconfirmed boxes with their float32 flags, detection
    boxes, column index lists, feasible edges, overrides):
      * the IoU cost matrix from the product's arithmetic equals the oracle's bit for bit (A13 once more, without a GPU);
      * certificate + replay through the box source equal the dense JV on the stage's full cost matrix, for body-only and
        face streams, dup-box scenes included."""
    import importlib
    import shutil
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "box"])
    lib = ctypes.CDLL(os.path.join(d, "liblapx_box_host.so"))
    P = ctypes.c_void_p
    lib.lapx_host_box_matrix.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, P, P]
    lib.lapx_host_exact_box.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                        ctypes.c_double, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    stages = tied = overrides = 0
    for kw in (dict(churn=0.1), dict(dup_frac=0.1, churn=0.05), dict(side_scale=40.0, dup_frac=0.05),
               dict(dup_frac=0.1, churn=0.05, face_dim=32)):
        st = pkg.synth.SyntheticStream(80, 40, 64, seed=17, **kw)
        ora = bo.TrackerOracle()
        for f in range(9):
            fr = st.next()
            ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None if fr.face_feats is None else fr.face_feats.copy())
            tr = ora.trace
            det = np.ascontiguousarray(tr["det_tlbr"], dtype=np.float64)
            for k, L, ab, af, bidx, iou_key in ((1, 0.8, "pool_tlbr", "pool_f32", "hi_idx", "iou1"), (2, 0.5, "r2_tlbr", "r2_f32", "lo_idx", "cost2"),
                                                (3, 0.7, "unc_tlbr", "unc_f32", "left_det", "iou3")):
                c = np.ascontiguousarray(tr["cost%d" % k], dtype=np.float64)
                n, m = c.shape
                if n == 0 or m == 0:
                    continue
                abox = np.ascontiguousarray(tr[ab], dtype=np.float64)
                af32 = np.ascontiguousarray(tr[af], dtype=np.uint8)
                cols = np.ascontiguousarray(tr[bidx], dtype=np.int32)
                assert abox.shape == (n, 4) and len(cols) == m
                got = np.empty((n, m), dtype=np.float64)
                lib.lapx_host_box_matrix(n, m, abox.ctypes.data, af32.ctypes.data, det.ctypes.data, cols.ctypes.data, got.ctypes.data)
                assert np.array_equal(got, np.asarray(tr[iou_key], dtype=np.float64)), (kw, f, k)      # the product's IoU arithmetic = the oracle's
                er, ec = np.nonzero(c <= L)
                orow, ocol = np.nonzero((c > L) & (c != got))
                E, O, hole = len(er), len(orow), 2
                e_row = np.concatenate([er, np.full(hole, -9), orow[::-1]]).astype(np.int32)
                e_col = np.concatenate([ec, np.full(hole, -9), ocol[::-1]]).astype(np.int32)
                e_cost = np.concatenate([c[er, ec], np.full(hole, -9.0), c[orow, ocol][::-1]]).astype(np.float64)
                want = jv(c, L)
                x0, y0 = scipy_visible(c, L)
                x, y, info = x0.copy(), y0.copy(), np.zeros(4, dtype=np.int32)
                rc = lib.lapx_host_exact_box(n, m, abox.ctypes.data, af32.ctypes.data, det.ctypes.data, cols.ctypes.data, e_row.ctypes.data,
                                             e_col.ctypes.data, e_cost.ctypes.data, E, O, E + hole, L, x.ctypes.data, y.ctypes.data, 64, f & 1,
                                             int(f % 3 == 0), info.ctypes.data)
                assert rc == 0 and info[1] == 0, (rc, info)
                assert np.array_equal(x, want[0]) and np.array_equal(y, want[1]), (kw, f, k, info)
                stages += 1; tied += int(info[0]); overrides += O
    assert stages > 70 and tied > 15 and overrides > 5, (stages, tied, overrides)
