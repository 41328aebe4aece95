"""The appearance pre-filter's exclusion rule, restated in NumPy and attacked on the CPU.

csrc/feat_math.h (to_i8_row, i8_excluded) turns every feature row into an int8 shadow plus records and lets the GEMM kernel drop
a (track, detection) pair -- after a PREFIX of the feature dimension or after the full row -- when

    dot_k / s^2 + alpha_a r_b + alpha_b r_a  <  uh_a + uh_b .

A dropped pair never gets its exact float32 dot product, so the rule has to be SOUND: dropped  =>  <a, b> < gate.  The
GPU suite checks that on the device's own shadows (test_prefilter_int8_shadows_bound_is_sound); this file checks the
MATHEMATICS (constants 1.03 / 68.7, the slack terms, the prefix energies) on families of rows chosen to break it, and that the
rule is not vacuous (it does drop what it is there to drop).  Records are built in float32 like the kernel's, the truth in
float64."""
import numpy as np
import pytest

I8_CLIP, I8_STAGE, NUM = 6.0, 128, (5, 7, 10, 13)


def checkpoints(kt_all):
    out, prev = [], 2
    for n in NUM:
        v = (kt_all * n + 15) // 16
        if v <= prev:
            v = prev + 1
        v = min(v, kt_all)
        out.append(v)
        prev = min(v, kt_all)
    return out


def shadows(x, gate):
    """x [n, D] float32 -> q int32 [n, D], records float32 [n, 5, 4] = {1/s, r, alpha_c, uh_c} (c = 4: the full row)."""
    x = np.asarray(x, dtype=np.float32)
    n, d = x.shape
    f = np.float32
    s = f(127.0) * np.sqrt(f(d)) / f(I8_CLIP)
    inv = f(1.0) / s
    q = np.rint(np.clip(x * s, -127.0, 127.0)).astype(np.int32)
    dq = q.astype(np.float32) * inv
    ss = (x * x).sum(1, dtype=np.float32)
    r = np.sqrt(((x - dq) ** 2).sum(1, dtype=np.float32)) * f(1.0001) + f(1e-5)
    cst = checkpoints(d // I8_STAGE)
    rec = np.zeros((n, 5, 4), dtype=np.float32)
    dq2 = dq * dq
    for c in range(5):
        k = d if c == 4 else cst[c] * I8_STAGE
        a_k = dq2[:, :k].sum(1, dtype=np.float32)
        if c == 4:
            alpha = np.maximum(np.sqrt(a_k), np.sqrt(ss)) * f(1.00001)
            uh = f(0.5) * f(gate) * np.ones(n, dtype=np.float32)
        else:
            alpha = np.zeros(n, dtype=np.float32)
            uh = f(0.5) * (a_k - f(1.03) * (ss - f(gate)) - f(68.7) * r * r)
        rec[:, c] = np.stack([np.full(n, inv, np.float32), r, alpha, uh - f(1e-4) * (f(1.0) + ss)], 1)
    return q, rec, cst


def track_side(rec, nrm, gate):
    """feat::i8_track_side: the record of a raw row x becomes the record of the unit row x / |x| a track holds."""
    f = np.float32
    out = rec.copy()
    k = np.asarray([1.03, 1.03, 1.03, 1.03, 1.0], dtype=np.float32)
    out[:, :, 3] = rec[:, :, 3] + k[None, :] * f(gate) * (nrm.astype(np.float32)[:, None] - f(1.0))
    bad = ~((nrm > 0) & (nrm < 3e38))
    out[bad, :, 3] = -np.inf
    return out


def excluded(qa, ra, qb, rb, c, cst, d):
    """Pairwise (row i of a with row i of b) at checkpoint c (4 = full row), float32 like i8_excluded."""
    k = d if c == 4 else cst[c] * I8_STAGE
    dot = (qa[:, :k].astype(np.int64) * qb[:, :k].astype(np.int64)).sum(1)
    A, B = ra[:, c], rb[:, c]
    lhs = dot.astype(np.float32) * A[:, 0] * B[:, 0] + (A[:, 2] * B[:, 1] + B[:, 2] * A[:, 1])
    return lhs < A[:, 3] + B[:, 3]


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def families(rng, n, d):
    """(name, a, b) row pairs; every family is built so that some pairs sit right at the gate."""
    base = unit(rng.standard_normal((n, d)))
    noise = unit(rng.standard_normal((n, d)))
    cosv = rng.uniform(0.55, 0.95, (n, 1))
    near = unit(cosv * base + np.sqrt(1 - cosv ** 2) * unit(noise - (noise * base).sum(1, keepdims=True) * base))
    yield "random impostors", base, unit(rng.standard_normal((n, d)))
    yield "near the gate", base, near
    # the energy of the common part sits entirely behind the first checkpoint / entirely in front of it
    k0 = checkpoints(d // I8_STAGE)[0] * I8_STAGE
    mask_late = np.zeros(d); mask_late[k0:] = 1.0
    for nm, m in (("agreement only in the tail", mask_late), ("agreement only in the head", 1.0 - mask_late)):
        if k0 >= d:                       # (rows too short for a prefix checkpoint: 384 = three stages)
            continue
        common = unit(rng.standard_normal((n, d)) * m)
        pa, pb = unit(rng.standard_normal((n, d)) * (1 - m)), unit(rng.standard_normal((n, d)) * (1 - m))
        w = rng.uniform(0.6, 0.99, (n, 1))
        yield nm, unit(np.sqrt(w) * common + np.sqrt(1 - w) * pa), unit(np.sqrt(w) * common + np.sqrt(1 - w) * pb)
    # spikes far beyond the clip level (6 / sqrt(D)): the quantiser saturates, the residual r carries the loss
    spiky = base.copy()
    idx = rng.integers(0, d, (n, 3))
    for j in range(3):
        spiky[np.arange(n), idx[:, j]] += rng.uniform(0.2, 0.7, n).astype(np.float32) * rng.choice([-1, 1], n)
    spiky = unit(spiky)
    yield "saturating spikes, same spikes", spiky, unit(spiky + 0.35 * unit(rng.standard_normal((n, d))))
    # rows that are not quite unit vectors (the reference renormalises in place only on first use, Q8) and tiny rows
    scale = rng.uniform(0.8, 1.25, (n, 1)).astype(np.float32)
    yield "norms 0.8 .. 1.25", (base * scale).astype(np.float32), (near * scale[::-1]).astype(np.float32)
    yield "one-hot-ish rows", unit(np.eye(d, dtype=np.float32)[rng.integers(0, d, n)] + 0.02 * base), \
        unit(np.eye(d, dtype=np.float32)[rng.integers(0, d, n)] + 0.02 * near)
    # non-negative (post-ReLU) embeddings: every pair has a large positive cosine
    yield "post-ReLU embeddings", unit(np.abs(base)), unit(np.abs(near))


@pytest.mark.parametrize("d", [384, 512, 2048])
def test_exclusion_is_sound(d):
    rng = np.random.default_rng(d)
    gate = 0.75
    total = dropped = 0
    for name, a, b in families(rng, 600, d):
        qa, ra, cst = shadows(a, gate)
        qb, rb, _ = shadows(b, gate)
        truth = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
        for c in range(5):
            if c < 4 and cst[c] * I8_STAGE >= d:
                continue
            ex = excluded(qa, ra, qb, rb, c, cst, d)
            bad = ex & (truth >= gate)
            assert not bad.any(), (name, d, c, float(truth[bad].max()), int(bad.sum()))
            total += len(ex); dropped += int(ex.sum())
    assert dropped > 0.15 * total         # (the families are mostly near-gate pairs: the rule still decides a good share)


def test_rule_is_not_vacuous_and_the_band_is_narrow():
    """What the rule is for: random impostors go at the FIRST checkpoint (5/16 of the row), and on the full row the band of
    undecided pairs below the gate is a few hundredths wide (the fp8 filter of round 1: 0.14)."""
    rng = np.random.default_rng(1)
    d, gate, n = 2048, 0.75, 1500
    a, b = unit(rng.standard_normal((n, d))), unit(rng.standard_normal((n, d)))
    qa, ra, cst = shadows(a, gate)
    qb, rb, _ = shadows(b, gate)
    assert cst == [5, 7, 10, 13]
    assert excluded(qa, ra, qb, rb, 0, cst, d).mean() > 0.99
    # pairs at controlled cosines: the full-row rule must drop everything below gate - 0.04 and keep everything above the gate
    for cos in (0.60, 0.68, 0.71):
        noise = unit(rng.standard_normal((n, d)))
        orth = unit(noise - (noise * a).sum(1, keepdims=True) * a)
        bb = unit(cos * a + np.sqrt(1 - cos ** 2) * orth)
        qb, rb, _ = shadows(bb, gate)
        assert excluded(qa, ra, qb, rb, 4, cst, d).mean() > 0.99, cos
    for cos in (0.76, 0.9):
        noise = unit(rng.standard_normal((n, d)))
        orth = unit(noise - (noise * a).sum(1, keepdims=True) * a)
        bb = unit(cos * a + np.sqrt(1 - cos ** 2) * orth)
        qb, rb, _ = shadows(bb, gate)
        for c in range(5):
            assert not excluded(qa, ra, qb, rb, c, cst, d).any(), (cos, c)


def test_non_finite_rows_are_never_dropped():
    rng = np.random.default_rng(2)
    d, gate = 512, 0.75
    a = unit(rng.standard_normal((8, d)))
    bad = a.copy(); bad[0, 3] = np.nan; bad[1, 7] = np.inf
    with np.errstate(invalid="ignore", over="ignore"):
        qa, ra, cst = shadows(bad, gate)
        qb, rb, _ = shadows(a, gate)
        # the kernel writes uh = -inf / r = +inf for such rows; the restatement's NaN records must not compare either
        ra[:2, :, 3] = -np.inf; ra[:2, :, 1] = np.inf
        for c in range(5):
            assert not excluded(qa, ra, qb, rb, c, cst, d)[:2].any()


@pytest.mark.parametrize("d", [384, 512, 2048])
def test_track_rows_copied_from_non_unit_detections_stay_sound(d):
    """a track's row is the RENORMALISED detection row x / |x| while its shadow is the shadow of the raw x.
    With the record turned into its track-side form (feat::i8_track_side, applied when the row is renormalised) the rule must
    stay sound for detection norms of 0.8 .. 1.25 on BOTH sides -- and the unadjusted record must be shown to fail there,
    so that this test cannot pass vacuously."""
    rng = np.random.default_rng(7 * d)
    gate, n = 0.75, 800
    broke_without = 0
    total = dropped = 0
    for lo, hi in ((0.8, 1.25), (0.8, 0.95), (1.05, 1.25), (1.0, 1.0)):
        base = unit(rng.standard_normal((n, d)))
        noise = unit(rng.standard_normal((n, d)))
        cosv = rng.uniform(0.6, 0.95, (n, 1))
        near = unit(cosv * base + np.sqrt(1 - cosv ** 2) * unit(noise - (noise * base).sum(1, keepdims=True) * base))
        na = rng.uniform(lo, hi, (n, 1)).astype(np.float32)
        nb = rng.uniform(lo, hi, (n, 1)).astype(np.float32)
        x_trk_raw = (base * na).astype(np.float32)            # the detection the track absorbed, as it arrived
        y_det = (near * nb).astype(np.float32)                # this frame's raw detection
        nrm = np.sqrt((x_trk_raw * x_trk_raw).sum(1, dtype=np.float32))
        u = (x_trk_raw / nrm[:, None]).astype(np.float32)     # what the track's curr feature holds
        truth = (u.astype(np.float64) * y_det.astype(np.float64)).sum(1)
        qa, ra_raw, cst = shadows(x_trk_raw, gate)
        qb, rb, _ = shadows(y_det, gate)
        ra = track_side(ra_raw, nrm, gate)
        for c in range(5):
            if c < 4 and cst[c] * I8_STAGE >= d:
                continue
            ex = excluded(qa, ra, qb, rb, c, cst, d)
            bad = ex & (truth >= gate)
            assert not bad.any(), (d, c, lo, hi, float(truth[bad].max()), int(bad.sum()))
            total += len(ex); dropped += int(ex.sum())
            broke_without += int((excluded(qa, ra_raw, qb, rb, c, cst, d) & (truth >= gate)).sum())
    assert dropped > 0.1 * total
    assert broke_without > 0, "the unadjusted record should drop true matches when |x| < 1 (the finding this test pins)"
    # degenerate norms: never dropped
    z = np.zeros((4, d), dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        qz, rz, cst = shadows(z, gate)
        rz = track_side(rz, np.zeros(4, dtype=np.float32), gate)
        qb, rb, _ = shadows(unit(rng.standard_normal((4, d))), gate)
        for c in range(5):
            assert not excluded(qz, rz, qb, rb, c, cst, d).any()


def test_restatement_matches_the_products_own_functions():
    """checkpoints(), the scale and track_side() above are NumPy restatements; the product's versions are __host__ __device__
    functions of csrc/feat_math.h -- called here on the CPU through tests/host_emul/libfeat_host.so (hipcc, host code only)."""
    import ctypes, os, shutil, subprocess
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "feat"])
    lib = ctypes.CDLL(os.path.join(d, "libfeat_host.so"))
    lib.feat_host_i8_scale.restype = ctypes.c_float
    lib.feat_host_track_side.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    assert lib.feat_host_meta_floats() == 20
    for dim in (128, 384, 512, 1024, 2048, 4096):
        out = (ctypes.c_int * 4)()
        lib.feat_host_checkpoints(dim // I8_STAGE, out)
        assert list(out) == checkpoints(dim // I8_STAGE), dim
        assert lib.feat_host_i8_scale(dim) == np.float32(127.0) * np.sqrt(np.float32(dim)) / np.float32(I8_CLIP)
    rng = np.random.default_rng(3)
    rec = rng.uniform(-1, 1, (64, 5, 4)).astype(np.float32)
    nrm = np.concatenate([rng.uniform(0.5, 1.5, 60), [0.0, np.inf, np.nan, 1.0]]).astype(np.float32)
    with np.errstate(invalid="ignore"):
        want = track_side(rec, nrm, 0.75)
    got = rec.copy()
    for i in range(64):
        row = np.ascontiguousarray(got[i].reshape(-1))
        lib.feat_host_track_side(row.ctypes.data, ctypes.c_float(nrm[i]), ctypes.c_float(0.75))
        got[i] = row.reshape(5, 4)
    assert np.array_equal(got, want, equal_nan=True)
