#!/usr/bin/env python
"""Where does k_filter_i8's time go?  Decodes the cycle accounting of a FILTER-TRACE build (scripts/build_variant.sh ftrace
FTRACE=1 -> ab/ftrace.so; run with BTS_LIB_PATH=ab/ftrace.so): per steady-state frame, from the kernel's own 100 MHz clock --
span of the GEMM workgroups (first start .. last end), how late the last persistent workgroup starts, the longest and the mean
workgroup lifetime, and inside a workgroup the mean time in candidate tests, waiting for operand stages, in the first test and in
the prologue before the first K stage; plus when the Kalman-predict workgroups of the same launch run (behind the GEMM by
default, in front with BTS_PREDICT_FIRST=1).  This is the decomposition of the "fixed 22.5 us" VERDICT r2 asks for."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "target"
    assert "ftrace" in pkg._lib.LIB_PATH, "run with BTS_LIB_PATH=<repo>/ab/ftrace.so (scripts/build_variant.sh ftrace FTRACE=1)"
    K, M, D = pkg.synth.workload_dims(wl)
    st = pkg.synth.make_workload(wl, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + max(64, K // 8), max_dets=max(K, M), feat_dim=D)
    rows = []
    for f in range(20):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
        torch.cuda.synchronize()
        trk.debug_filter()
        r = trk._filter_raw.astype(np.int64)
        if f >= 10 and r[1] > 0:
            rows.append(r)
    us = 0.01                                              # one tick of s_memrealtime = 10 ns
    print("%s, %s predict workgroups %s the GEMM" % (wl, "BTS_PREDICT_FIRST=1:" if os.environ.get("BTS_PREDICT_FIRST") else "default:",
                                                       "in front of" if os.environ.get("BTS_PREDICT_FIRST") else "behind"))
    print("frame  tiles stages | gemm span  last start  longest wg  mean wg | tests  stage waits  first test  prologue | predict window (from the first GEMM start)")
    for k, r in enumerate(rows):
        nwg = int(min(r[1], 512))
        span, late, longest = (r[9] - r[8]) * us, (r[11] - r[8]) * us, r[10] * us
        mean = lambda v: v * us / max(nwg, 1)
        print("%5d %6d %6d | %8.1f %10.1f %11.1f %8.1f | %5.1f %11.1f %10.1f %9.1f | %6.1f .. %6.1f us"
              % (k, r[1], r[0], span, late, longest, mean(r[7]), mean(r[5]), mean(r[6]), mean(r[3]), mean(r[4]),
                 (r[12] - r[8]) * us, (r[13] - r[8]) * us))
    trk.close()


if __name__ == "__main__":
    main()
