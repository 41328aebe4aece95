"""The product's Kalman and cost arithmetic on the CPU against the REFERENCE's golden vectors.

csrc/kf_math.h's scalar forms (namespace kf: what k_initiate / k_project / k_gating / k_gmc, the tracker's births and the motion
gate execute per thread, and what the 8-lane forms of predict / update are written to reproduce) are __host__ __device__;
tests/host_emul/kf_host.hip compiles them with hipcc for the host.  Same bars as the GPU tests (tests/test_gpu_kernels.py):
initiate / predict / project bit-identical to the vectors recorded from the reference (G1), update <= 1e-9 relative (LAPACK's
summation order differs), gating <= 1e-10, multi_gmc <= 1e-12 (G7).  Round 3 had no GPU: this is the parity evidence for rows
A2 - A8 / A22 that does not need one (the machine code of every kernel is unchanged since the last GPU-green run,
profiles/r03_isa_vs_8cbc337.txt)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import golden

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kf():
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "kf"])
    lib = ctypes.CDLL(os.path.join(d, "libkf_host.so"))
    P = ctypes.c_void_p
    for name, args in (("kf_host_initiate", [ctypes.c_int, P, P, P]), ("kf_host_predict", [ctypes.c_int, P, P, P, P]),
                       ("kf_host_project", [ctypes.c_int, P, P, P, P, P]), ("kf_host_update", [ctypes.c_int, P, P, P, P]),
                       ("kf_host_tlbr", [ctypes.c_int, P, P, P]), ("kf_host_gmc", [ctypes.c_int, P, P, P]),
                       ("kf_host_maha", [P, P, ctypes.c_int, P, P])):
        getattr(lib, name).argtypes = args
    return lib


def c64(a):
    return np.ascontiguousarray(a, dtype=np.float64).copy()


def close(a, b, rel=1e-9, abs_=1e-12):
    err = np.abs(a - b)
    tol = abs_ + rel * np.maximum(np.abs(a), np.abs(b))
    assert (err <= tol).all(), float((err - tol).max())


def test_initiate_bitwise(kf):
    g = golden("g1_kalman.npz")
    z = np.ascontiguousarray(g["meas32"], dtype=np.float32)
    n = len(z)
    m, c = np.empty((n, 8)), np.empty((n, 8, 8))
    kf.kf_host_initiate(n, z.ctypes.data, m.ctypes.data, c.ctypes.data)
    # the reference's state is float32 here (NumPy 2 scalar promotion); the product carries float64 holders of the same values
    assert np.array_equal(m, g["init_mean"].astype(np.float64)) and np.array_equal(c, g["init_cov"].astype(np.float64))


def test_predict_bitwise(kf):
    g = golden("g1_kalman.npz")
    n = len(g["state_mean"])
    m, c = c64(g["state_mean"]), c64(g["state_cov"])
    kf.kf_host_predict(n, m.ctypes.data, c.ctypes.data, None, None)
    assert np.array_equal(m, g["multi_predict_mean"]) and np.array_equal(c, g["multi_predict_cov"])
    assert np.array_equal(m[3], g["predict_mean"][3]) and np.array_equal(c[3], g["predict_cov"][3])
    m, c = c64(g["state_mean"]), c64(g["state_cov"])
    zv = np.ascontiguousarray(g["not_tracked"], dtype=np.uint8)
    kf.kf_host_predict(n, m.ctypes.data, c.ctypes.data, None, zv.ctypes.data)             # the vw, vh = 0 rule (A8)
    assert np.array_equal(m, g["multi_predict_z_mean"]) and np.array_equal(c, g["multi_predict_z_cov"])


def test_predict_of_a_float32_batch(kf):
    """frame 2: every pool row still holds the float32 state of initiate(): NumPy evaluates the process noise in float32 (Q10)."""
    from oracle import bts_oracle as bo
    g = golden("g1_kalman.npz")
    want_m, want_c = bo.KalmanOracle().multi_predict(g["init_mean"].copy(), g["init_cov"].copy())
    n = len(want_m)
    m, c = c64(g["init_mean"]), c64(g["init_cov"])
    ones = np.ones(n, dtype=np.uint8)
    kf.kf_host_predict(n, m.ctypes.data, c.ctypes.data, ones.ctypes.data, None)
    assert np.array_equal(m, want_m.astype(np.float64)) and np.array_equal(c, want_c.astype(np.float64))


def test_project_bitwise_and_tlbr(kf):
    g = golden("g1_kalman.npz")
    n = len(g["state_mean"])
    m, c = c64(g["state_mean"]), c64(g["state_cov"])
    z, s = np.empty((n, 4)), np.empty((n, 4, 4))
    kf.kf_host_project(n, m.ctypes.data, c.ctypes.data, None, z.ctypes.data, s.ctypes.data)
    assert np.array_equal(z, g["project_mean"]) and np.array_equal(s, g["project_cov"])
    out = np.empty((n, 4))
    kf.kf_host_tlbr(n, m.ctypes.data, None, out.ctypes.data)
    want = m[:, :4].copy(); want[:, :2] -= want[:, 2:] / 2; want[:, 2:] += want[:, :2]        # D:624-642
    assert np.array_equal(out, want)


def test_update_vs_golden(kf):
    g = golden("g1_kalman.npz")
    n = len(g["multi_predict_mean"])
    m, c = c64(g["multi_predict_mean"]), c64(g["multi_predict_cov"])
    z = np.ascontiguousarray(g["update_z"], dtype=np.float32)
    kf.kf_host_update(n, m.ctypes.data, c.ctypes.data, z.ctypes.data, None)
    close(m, g["update_mean"])
    close(c, g["update_cov"], rel=1e-9, abs_=1e-9)


def test_update_of_float32_state_vs_oracle(kf):
    from oracle import bts_oracle as bo
    g = golden("g1_kalman.npz")
    ko = bo.KalmanOracle()
    for i in range(6):
        m0, c0 = g["init_mean"][i], g["init_cov"][i]
        z = (m0[:4] + np.float32(1.5)).astype(np.float32)
        wm, wc = ko.update(m0, c0, z)
        m, c = c64(m0), c64(c0)
        one = np.ones(1, dtype=np.uint8)
        kf.kf_host_update(1, m.ctypes.data, c.ctypes.data, z.ctypes.data, one.ctypes.data)
        close(m, np.asarray(wm, dtype=np.float64))
        close(c, np.asarray(wc, dtype=np.float64), abs_=1e-9)


def test_gating_distance(kf):
    g = golden("g1_kalman.npz")
    meas = c64(g["gate_meas"])
    for i in range(4):
        m, c = c64(g["state_mean"][i]), c64(g["state_cov"][i])
        out = np.empty(len(meas))
        kf.kf_host_maha(m.ctypes.data, c.ctypes.data, len(meas), meas.ctypes.data, out.ctypes.data)
        close(out, g["gate_%d_0_maha" % i], rel=1e-10)


def test_multi_gmc(kf):
    g, g1 = golden("g7_gmc.npz"), golden("g1_kalman.npz")
    n = len(g1["state_mean"])
    m, c = c64(g1["state_mean"]), c64(g1["state_cov"])
    h = c64(g["H"]).reshape(6)
    kf.kf_host_gmc(n, m.ctypes.data, c.ctypes.data, h.ctypes.data)
    close(m, g["mean"], rel=1e-13, abs_=1e-12)
    close(c, g["cov"], rel=1e-12, abs_=1e-12)


# ---- the 8-lane forms (kf8::): the code of the tracker's predict and update paths ------------------------------------------
def _kf8(kf):
    P = ctypes.c_void_p
    kf.kf8_host_predict.argtypes = [ctypes.c_int, P, P, P, P]
    kf.kf8_host_update.argtypes = [ctypes.c_int, P, P, P, P]
    kf.kf8_host_tlbr.argtypes = [ctypes.c_int, P, P, P]
    return kf


def test_eight_lane_predict_bitwise(kf):
    """kf8::predict -- eight lanes per track trading values through __shfl on the device, run in lockstep by replay here -- is
    bit-identical to the reference's multi_predict (G1), incl. the vw, vh = 0 rule and the all-float32 batch (Q10), and to the
    scalar form."""
    from oracle import bts_oracle as bo
    _kf8(kf)
    g = golden("g1_kalman.npz")
    n = len(g["state_mean"])
    m, c = c64(g["state_mean"]), c64(g["state_cov"])
    kf.kf8_host_predict(n, m.ctypes.data, c.ctypes.data, None, None)
    assert np.array_equal(m, g["multi_predict_mean"]) and np.array_equal(c, g["multi_predict_cov"])
    m, c = c64(g["state_mean"]), c64(g["state_cov"])
    zv = np.ascontiguousarray(g["not_tracked"], dtype=np.uint8)
    kf.kf8_host_predict(n, m.ctypes.data, c.ctypes.data, None, zv.ctypes.data)
    assert np.array_equal(m, g["multi_predict_z_mean"]) and np.array_equal(c, g["multi_predict_z_cov"])
    want_m, want_c = bo.KalmanOracle().multi_predict(g["init_mean"].copy(), g["init_cov"].copy())
    m, c = c64(g["init_mean"]), c64(g["init_cov"])
    ones = np.ones(n, dtype=np.uint8)
    kf.kf8_host_predict(n, m.ctypes.data, c.ctypes.data, ones.ctypes.data, None)
    assert np.array_equal(m, want_m.astype(np.float64)) and np.array_equal(c, want_c.astype(np.float64))
    out = np.empty((n, 4))
    kf.kf8_host_tlbr(n, m.ctypes.data, None, out.ctypes.data)
    want = m[:, :4].copy(); want[:, :2] -= want[:, 2:] / 2; want[:, 2:] += want[:, :2]
    assert np.array_equal(out, want)


def test_eight_lane_update_vs_golden_and_scalar_form(kf):
    _kf8(kf)
    g = golden("g1_kalman.npz")
    n = len(g["multi_predict_mean"])
    z = np.ascontiguousarray(g["update_z"], dtype=np.float32)
    m, c = c64(g["multi_predict_mean"]), c64(g["multi_predict_cov"])
    kf.kf8_host_update(n, m.ctypes.data, c.ctypes.data, z.ctypes.data, None)
    close(m, g["update_mean"])
    close(c, g["update_cov"], rel=1e-9, abs_=1e-9)
    ms, cs = c64(g["multi_predict_mean"]), c64(g["multi_predict_cov"])
    kf.kf_host_update(n, ms.ctypes.data, cs.ctypes.data, z.ctypes.data, None)
    assert np.array_equal(m, ms) and np.array_equal(c, cs)            # "the same sequence of operations as kf::update"
    # float32 state (unconfirmed tracks updated straight from initiate(), Q5 / Q10)
    m, c = c64(g["init_mean"]), c64(g["init_cov"])
    ms, cs = m.copy(), c.copy()
    z2 = np.ascontiguousarray(g["init_mean"][:, :4] + np.float32(1.5), dtype=np.float32)
    ones = np.ones(n, dtype=np.uint8)
    kf.kf8_host_update(n, m.ctypes.data, c.ctypes.data, z2.ctypes.data, ones.ctypes.data)
    kf.kf_host_update(n, ms.ctypes.data, cs.ctypes.data, z2.ctypes.data, ones.ctypes.data)
    assert np.array_equal(m, ms) and np.array_equal(c, cs)


# ---- the fused association costs (csrc/cost_math.h) -------------------------------------------------------------------------
def test_fused_costs_on_the_cpu_vs_golden():
    """fuse1 / fuse3 / face_fixup -- what k_fuse1 / k_fuse3 evaluate per element -- compiled for the host: stage-1 costs
    bit-identical to the matrices recorded from the reference (G3), and to the oracle on values exactly at the thresholds and at
    the face "0.9999999" rule."""
    from oracle import bts_oracle as bo
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "cost"])
    lib = ctypes.CDLL(os.path.join(d, "libcost_host.so"))
    P = ctypes.c_void_p
    lib.cost_host_fuse1.argtypes = [ctypes.c_long, P, P, P, ctypes.c_double, ctypes.c_float, P]
    lib.cost_host_fuse3.argtypes = [ctypes.c_long, P, P, ctypes.c_double, ctypes.c_float, P]

    def f1(iou_d, body, face):
        i, b = c64(iou_d), np.ascontiguousarray(body, dtype=np.float32)
        f = None if face is None else np.ascontiguousarray(face, dtype=np.float32)
        out = np.empty_like(i)
        lib.cost_host_fuse1(i.size, i.ctypes.data, b.ctypes.data, None if f is None else f.ctypes.data, 0.5, 0.25, out.ctypes.data)
        return out

    def f3(iou_d, body):
        i, b = c64(iou_d), np.ascontiguousarray(body, dtype=np.float32)
        out = np.empty_like(i)
        lib.cost_host_fuse3(i.size, i.ctypes.data, b.ctypes.data, 0.5, 0.25, out.ctypes.data)
        return out

    g = golden("g3_fused.npz")
    for k in range(int(g["n_rec"])):
        assert np.array_equal(f1(g["s1_%d_iou" % k], g["s1_%d_body_sim" % k], g["s1_%d_face_sim" % k]), g["s1_%d_cost" % k]), k
        if "s3_%d_cost" % k in g.files:
            sim = bo.cosine_similarity(g["s3_%d_trk_feats" % k], g["s3_%d_det_feats" % k])       # (the GEMM itself needs the GPU)
            assert np.abs(f3(g["s3_%d_iou" % k], sim) - g["s3_%d_cost" % k]).max() <= 2e-6, k
    rng = np.random.default_rng(9)
    iou_d = rng.uniform(0, 1, (40, 33))
    body = rng.uniform(0, 1, (40, 33)).astype(np.float32)
    face = rng.uniform(0, 1, (40, 33)).astype(np.float32)
    body[0, :4] = [0.75, np.float32(0.75) + np.float32(1e-7), 0.7499999, 1.0]
    face[1, :3] = [0.9999999, np.float32(1.0), 0.99999994]
    iou_d[2, :3] = [0.5, 0.5000000001, 0.4999999]
    assert np.array_equal(f1(iou_d, body, face), bo.fuse_cost_stage1(iou_d, body.copy(), bo.face_similarity_fixup(face)))
    assert np.array_equal(f3(iou_d, body), bo.fuse_cost_stage3(iou_d, body.copy()))
    assert np.array_equal(f1(iou_d, body, None), bo.fuse_cost_stage1(iou_d, body.copy(), np.zeros_like(face)))


def test_overlap_screening_never_misses_a_pair():
    """The cost and duplicate tiles skip a pair's exact float64 IoU when iou::screen_overlap says the boxes are disjoint (boxes
    rounded OUTWARD to 16-bit pixels, packed compare).  A missed overlap would be a missing edge -- a silent parity bug -- so the
    screening must say "maybe" for EVERY pair whose exact boxes overlap: fractional coordinates, boxes touching at a pixel
    boundary, sub-pixel boxes, coordinates far outside the 16-bit range, negative coordinates.  (And it must not be vacuous.)"""
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "box"])
    lib = ctypes.CDLL(os.path.join(d, "liblapx_box_host.so"))
    P = ctypes.c_void_p
    lib.iou_host_screen.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P]
    lib.iou_host_screen_none.argtypes = [ctypes.c_int, P]
    rng = np.random.default_rng(21)

    def boxes(n, lo, hi, wmax, frac=True):
        x = rng.uniform(lo, hi, n); y = rng.uniform(lo, hi, n)
        w = rng.uniform(0.01, wmax, n); h = rng.uniform(0.01, wmax, n)
        b = np.stack([x, y, x + w, y + h], 1)
        return b if frac else np.rint(b)

    skipped = total = 0
    for lo, hi, wmax, frac in ((0, 2000, 150, True), (0, 300, 40, False), (-500, 500, 90, True), (15900, 16100, 50, True),
                               (-16100, -15900, 50, True), (-1e6, 1e6, 3e5, True), (0, 50, 0.9, True), (1e9, 3e9, 1e9, True)):
        a, b = boxes(300, lo, hi, wmax, frac), boxes(260, lo, hi, wmax, frac)
        a[:20, 2:] = a[:20, :2] + np.rint(rng.uniform(1, 30, (20, 2)))                   # integer-sized boxes ...
        b[:20, :2] = a[:20, 2:]                                                          # ... whose neighbour starts exactly where they end (touching: no overlap)
        b[:20, 2:] = b[:20, :2] + 10
        b[20:40] = a[20:40]                                                              # identical boxes
        out = np.zeros((len(a), len(b)), dtype=np.uint8)
        lib.iou_host_screen(len(a), len(b), np.ascontiguousarray(a).ctypes.data, np.ascontiguousarray(b).ctypes.data, out.ctypes.data)
        x_lo = np.maximum(a[:, None, 0], b[None, :, 0]); x_hi = np.minimum(a[:, None, 2], b[None, :, 2])
        y_lo = np.maximum(a[:, None, 1], b[None, :, 1]); y_hi = np.minimum(a[:, None, 3], b[None, :, 3])
        overlap = (x_hi > x_lo) & (y_hi > y_lo)                                          # D:1705: anything else is IoU 0 exactly
        assert not (overlap & (out == 0)).any(), (lo, hi, int((overlap & (out == 0)).sum()))
        skipped += int((out == 0).sum()); total += out.size
        assert lib.iou_host_screen_none(len(b), np.ascontiguousarray(b).ctypes.data) == 0
    assert skipped > 0.5 * total          # most disjoint pairs are indeed screened out
    nan = np.full((1, 4), np.nan)
    out = np.zeros((1, 1), dtype=np.uint8)
    lib.iou_host_screen(1, 1, nan.ctypes.data, np.asarray([[0.0, 0.0, 5.0, 5.0]]).ctypes.data, out.ctypes.data)      # (must not crash)


def test_cost_rounding_is_numpys_float16():
    """bts_tracker_config.cost_dtype = BTS_F16 (BASELINE config 5's "fp16 cost matrix") rounds every association cost with
    csrc/cost_math.h round_to_f16: that IS float64 -> binary16 -> float64 as NumPy does it (round to nearest even, straight
    from float64) -- normals, ties exactly between two half values, subnormals, overflow, signs, inf / NaN."""
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    d = os.path.join(HERE, "host_emul")
    subprocess.check_call(["make", "-C", d, "-s", "cost"])
    lib = ctypes.CDLL(os.path.join(d, "libcost_host.so"))
    lib.cost_host_round_f16.argtypes = [ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(16)
    halves = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float64)       # every finite non-negative half
    mids = (halves[:-1] + halves[1:]) / 2                                                     # exact ties
    v = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(0, 1, 20000) ** 8, halves, mids, np.nextafter(mids, 0), np.nextafter(mids, 1e9),
                        10.0 ** rng.uniform(-9, 5.2, 50000), [0.0, 65504.0, 65519.999, 65520.0, 1e300, np.inf, 5.96e-8, 2.98e-8, 2.9802322387695312e-08, 1e-12]])
    v = np.ascontiguousarray(np.concatenate([v, -v, [np.nan]]))
    out = np.empty_like(v)
    lib.cost_host_round_f16(v.size, v.ctypes.data, out.ctypes.data)
    with np.errstate(over="ignore"):
        want = v.astype(np.float16).astype(np.float64)
    assert np.array_equal(out, want, equal_nan=True), np.flatnonzero(~((out == want) | (np.isnan(out) & np.isnan(want))))[:10]
    assert np.array_equal(np.signbit(out), np.signbit(want))
