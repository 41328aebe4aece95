#!/usr/bin/env python
"""Serial step counts of the assignment kernel's component solvers on a workload's stage-1 problems -- measured on the CPU by
running the product's k_lap through the HIP emulator (tests/hip_emu) with its emulator-only counters (LSTAT in lap.hip).
A component is solved by ONE wavefront, so the time of the launch is bounded from below by the steps of its largest
component: steps are hardware-independent, which makes this the one k_lap figure that can be improved without a GPU.

    BTS_LIB_PATH=tests/hip_emu/libbotsort_emu.so python scripts/lap_steps_cpu.py [--workload crowd] [--frames 4]
"""
import argparse
import ctypes
import importlib
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("BTS_LIB_PATH", os.path.join(REPO, "tests", "hip_emu", "libbotsort_emu.so"))
import tests.hip_emu.emu_prelude                                      # noqa: E402,F401
from oracle import bts_oracle as bo                                   # noqa: E402

pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
NAMES = ["packed", "tall", "global", "aug", "steps", "relax", "tall_aug", "tall_steps", "tall_relax", "max_comp_steps", "path", "tall_prelude", "big_or_global_rows", "big_or_global_cols", "big_or_global_steps", "big", "prep_lds_a", "prep_lds_b", "prep_global", "prep_started_over"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="crowd")
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--M", type=int, default=1024)
    ap.add_argument("--D", type=int, default=64, help="feature width (geometry decides the components; small keeps the oracle fast)")
    ap.add_argument("--frames", type=int, default=4)
    a = ap.parse_args()
    lib = ctypes.CDLL(os.environ["BTS_LIB_PATH"])
    lib.bts_emu_lap_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    st = pkg.synth.SyntheticStream(a.K, a.M, a.D, seed=11, **pkg.synth.ADVERSARIAL.get(a.workload, {}))
    ora = bo.TrackerOracle()
    tot = np.zeros(24, dtype=np.int64)
    for f in range(a.frames + 1):
        fr = st.next()
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        c = ora.trace["cost1"]
        if c.size == 0:
            continue
        lib.bts_emu_lap_stats(None, 1)
        t0 = time.time()
        _, x, y = pkg.matching.lapjv(c, extend_cost=True, cost_limit=0.8)
        out = np.zeros(24, dtype=np.int64)
        lib.bts_emu_lap_stats(out.ctypes.data, 1)
        want = ora.trace["m1"]
        got = np.asarray([[i, j] for i, j in enumerate(x) if j >= 0]).reshape(-1, 2)
        ok = np.array_equal(got, np.asarray(want).reshape(-1, 2))
        print("frame %d  %dx%d  edges %d  == oracle: %s  (%.0f s)  " % (f, c.shape[0], c.shape[1], int((c <= 0.8).sum()), ok, time.time() - t0)
              + "  ".join("%s %d" % (n, v) for n, v in zip(NAMES, out) if v))
        tot += out
    print("total  " + "  ".join("%s %d" % (n, v) for n, v in zip(NAMES, tot) if v))


if __name__ == "__main__":
    main()
