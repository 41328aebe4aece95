#!/usr/bin/env python
"""Where do exact cost ties occur, and what do they do?  CPU survey (no GPU needed) with the product's exact-tie source
compiled for the host (tests/host_emul) next to the oracle's dense JV (= lap.lapjv):

  1. per tie family: how often the uniqueness certificate says "tied", how often SOME optimum (SciPy's Hungarian on the
     extended matrix: a stand-in for any exact solver without lap's scan orders) differs from the dense JV's indices, and
     that certificate + replay always ends at the dense JV's indices;
  2. per synthetic workload: in how many association stages of a stream the certificate fires at all (plain streams: never;
     the dup-box scene: wherever a duplicated box is in play).
Output kept as profiles/r03_tie_survey_cpu.txt.  The device-side twin (fast solver vs dense JV, indices) is
scripts/tie_survey.py / tests/test_zz_gpu_ties.py."""
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import bts_oracle as bo                                   # noqa: E402
from tests import test_lap_exact_host as T                            # noqa: E402

pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")


def host_runner():
    import ctypes, subprocess
    d = os.path.join(REPO, "tests", "host_emul")
    subprocess.check_call(["make", "-C", d, "-s"])
    lib = ctypes.CDLL(os.path.join(d, "liblapx_host.so"))
    P = ctypes.c_void_p
    lib.lapx_host_exact.argtypes = [ctypes.c_int, ctypes.c_int, P, ctypes.c_int, ctypes.c_double, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_uint, P]

    def f(c, L, x, y):
        c = np.ascontiguousarray(c, dtype=np.float64); n, m = c.shape
        x = np.ascontiguousarray(x, dtype=np.int32).copy(); y = np.ascontiguousarray(y, dtype=np.int32).copy()
        info = np.zeros(4, dtype=np.int32)
        rc = lib.lapx_host_exact(n, m, c.ctypes.data, m, L, x.ctypes.data, y.ctypes.data, 64, 0, 0, None, 0, info.ctypes.data)
        assert rc == 0 and info[1] == 0
        return x, y, bool(info[0])
    return f


def family(name, cases, L, f):
    tied = differs = final_ok = 0
    for c in cases:
        c = np.ascontiguousarray(c, dtype=np.float64)
        want = T.jv(c, L)
        x0, y0 = T.scipy_visible(c, L)
        x, y, t = f(c, L, x0, y0)
        tied += t
        differs += not (np.array_equal(x0, want[0]) and np.array_equal(y0, want[1]))
        final_ok += np.array_equal(x, want[0]) and np.array_equal(y, want[1])
    print("%-36s cases %4d  certificate 'tied' %4d  some-optimum != dense JV %4d  certificate+replay == dense JV %4d"
          % (name, len(cases), tied, differs, final_ok))


def workload(name, K, M, D, frames, f, **kw):
    st = pkg.synth.SyntheticStream(K, M, D, seed=77, **kw)
    ora = bo.TrackerOracle()
    stages = tied = differs = 0
    for _ in range(frames):
        fr = st.next()
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        for k, L in ((1, 0.8), (2, 0.5), (3, 0.7)):
            c = ora.trace["cost%d" % k]
            if c.size == 0:
                continue
            want = T.jv(c, L)
            x0, y0 = T.scipy_visible(c, L)
            _, _, t = f(c, L, x0, y0)
            stages += 1; tied += t
            differs += not (np.array_equal(x0, want[0]) and np.array_equal(y0, want[1]))
    print("%-36s stages %4d  certificate 'tied' %4d  some-optimum != dense JV %4d" % (name, stages, tied, differs))


def main():
    f = host_runner()
    rng = np.random.default_rng(0)
    L = 0.8
    family("1 row x 2 equal columns", [np.full((1, 2), v) for v in (0.1, 0.3, 0.5, 0.79)], L, f)
    family("2 rows x 1 equal column", [np.full((2, 1), v) for v in (0.1, 0.3, 0.5, 0.79)], L, f)
    family("k x k all-equal blocks", [np.full((k, k), 0.25) for k in range(2, 10)], L, f)
    family("n x m all-equal blocks", [np.full((n, m), 0.25) for n in range(1, 6) for m in range(1, 6) if n != m], L, f)
    family("cost exactly at the limit", [np.full((1, 1), L), np.array([[L, 0.2], [0.3, L]]), np.array([[L, L], [L, L]])], L, f)
    family("cost exactly at half the limit", [np.full((1, 1), L / 2), np.array([[L / 2, L / 2]]), np.array([[L / 2], [L / 2]])], L, f)
    cases = []
    for _ in range(200):
        n, m = rng.integers(2, 9, 2)
        cases.append(rng.integers(1, 5, (n, m)) / 5.0)
    family("small integer-grid costs", cases, L, f)
    cases = []
    for _ in range(100):
        n, m = rng.integers(3, 12, 2)
        c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
        j = rng.integers(0, m)
        cases.append(np.concatenate([c, c[:, j:j + 1]], axis=1))
    family("random + one duplicated column", cases, L, f)
    cases = []
    for _ in range(100):
        n, m = rng.integers(3, 12, 2)
        c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
        i = rng.integers(0, n)
        cases.append(np.concatenate([c, c[i:i + 1]], axis=0))
    family("random + one duplicated row", cases, L, f)
    cases = []
    for _ in range(300):
        n, m = rng.integers(3, 40, 2)
        c = rng.uniform(0.05, 0.95, (n, m)); c[rng.random((n, m)) < 0.7] = 1.0
        cases.append(c)
    family("tie-free random float costs", cases, L, f)
    print()
    workload("c1 64x32x128 plain stream", 64, 32, 128, 20, f)
    workload("256x128x128 plain, churn 0.1", 256, 128, 128, 12, f, churn=0.1)
    workload("crowd 256x128x128 (boxes 5-10 deep)", 256, 128, 128, 8, f, side_scale=40.0)
    workload("dup-box 64x32x128 (10 % copies)", 64, 32, 128, 20, f, dup_frac=0.1, churn=0.05)
    workload("dup-box 256x128x128 (5 % copies)", 256, 128, 128, 12, f, dup_frac=0.05, churn=0.05)


if __name__ == "__main__":
    main()
