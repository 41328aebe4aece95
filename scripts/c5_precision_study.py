#!/usr/bin/env python
"""BASELINE config 5 names an arithmetic mode -- "fp32 Kalman / fp16 cost matrix" -- that this implementation deliberately does
not have (Kalman state float64, edge costs float64; there is no cost matrix to store).  What WOULD that mode
do to the metric's "assignment ID match vs ref"?  CPU study with the oracle (no GPU needed): the reference-faithful float64
tracker next to the same tracker with (a) every Kalman operation in float32 and (b) every stage's cost matrix rounded to
float16 before the assignment, on the same seeded streams.  Reports, per stream: frames whose three assignments and track
lists are identical, first frame of divergence, largest relative difference of a Kalman mean while the lists still agree.
Output kept as profiles/r03_c5_precision_study.txt."""
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import bts_oracle as bo                                   # noqa: E402

pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")


class KalmanF32(bo.KalmanOracle):
    """Every operation on float32 operands (NumPy / LAPACK single precision), results float32."""

    def __init__(self):
        super().__init__()
        self.F = self.F.astype(np.float32); self.H = self.H.astype(np.float32)
        self.w_pos = np.float32(self.w_pos); self.w_vel = np.float32(self.w_vel)

    @staticmethod
    def _f(*a):
        return [np.asarray(v, dtype=np.float32) for v in a]

    def initiate(self, z):
        m, c = super().initiate(np.asarray(z, dtype=np.float32))
        return self._f(m, c)

    def multi_predict(self, mean, cov):
        return self._f(*super().multi_predict(*self._f(mean, cov)))

    def update(self, mean, cov, z):
        return self._f(*super().update(*self._f(mean, cov, z)))


class LowPrecisionTracker(bo.TrackerOracle):
    def __init__(self, kalman32, cost16, **kw):
        super().__init__(**kw)
        self.cost16 = cost16
        if kalman32:
            self.kf = KalmanF32()


def run(name, K, M, D, frames, kalman32, cost16, **kw):
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234, **kw)
    ref, low = bo.TrackerOracle(), LowPrecisionTracker(kalman32, cost16)
    orig = bo.linear_assignment
    same = 0
    first_div = None
    worst = 0.0
    for f in range(frames):
        fr = st.next()
        ref.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        if cost16:
            bo.linear_assignment = lambda c, t: orig(np.asarray(c).astype(np.float16).astype(np.float64), float(np.float16(t)))
        try:
            low.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        finally:
            bo.linear_assignment = orig
        ok = all(np.array_equal(ref.trace["m%d" % k], low.trace["m%d" % k]) for k in (1, 2, 3))
        ok = ok and [t.track_id for t in ref.tracked] == [t.track_id for t in low.tracked] and [t.track_id for t in ref.lost] == [t.track_id for t in low.lost]
        if ok:
            same += 1
            for a, b in zip(ref.tracked, low.tracked):
                ma, mb = np.asarray(a.mean, dtype=np.float64), np.asarray(b.mean, dtype=np.float64)
                worst = max(worst, float(np.max(np.abs(ma[:4] - mb[:4]) / np.maximum(1.0, np.abs(ma[:4])))))
        elif first_div is None:
            first_div = f
    print("%-34s kalman %-3s cost %-4s: %3d / %3d frames identical, first divergence %-5s max rel. mean diff (position, while identical) %.2e"
          % (name, "f32" if kalman32 else "f64", "f16" if cost16 else "f64", same, frames, "-" if first_div is None else "f%d" % first_div, worst))


def main():
    for name, K, M, D, frames, kw in (("c2 512x256x128", 512, 256, 128, 40, {}),
                                      ("1536x384x384 (c5 shape / 10)", 1536, 384, 384, 25, {}),
                                      ("crowd 512x256x128", 512, 256, 128, 25, dict(side_scale=40.0)),
                                      ("churn 512x256x128", 512, 256, 128, 40, dict(churn=0.1))):
        for k32, c16 in ((True, False), (False, True), (True, True)):
            run(name, K, M, D, frames, k32, c16, **kw)
        print()


if __name__ == "__main__":
    main()
