#!/usr/bin/env python
"""Digest-equality soak of BTS_CVT_OVERLAP (the side-stream conversion of the detection rows): the same
frames through a default tracker and one created with BTS_CVT_OVERLAP=1, every frame's digest compared (assignment vectors of
the three stages, both lists in order, means, boxes, scores: tune._digest), with and without the residency promise
(bts_tracker_inputs_resident), and -- to give a race a chance -- WITHOUT a host synchronisation between the steps of the
overlapped tracker: its digests are taken every `stride` frames only (a digest synchronises), the default tracker's at the same
frames.  The form's two events per frame are all that orders its side stream against the step's stream; the CPU emulator cannot
see them (events are no-ops there), so this soak on the device is the form's evidence.  tune.py offers the form only under
BTS_TUNE_ALLOW_CVT=1, to be set once this has passed.

Round 6: the same soak serves BTS_FILTER_AHEAD (the pre-filter of frame f on the side stream, ahead of its step) -- same kind of
evidence, same gate:    python scripts/soak_cvt_overlap.py [frames=2000] [forms, default "BTS_CVT_OVERLAP=1,BTS_LISTS_AHEAD=1"]

    python scripts/soak_cvt_overlap.py [frames=2000]
    python scripts/soak_cvt_overlap.py 2000 BTS_FILTER_AHEAD=1"""
import importlib
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")


FORMS = sys.argv[2] if len(sys.argv) > 2 else "BTS_CVT_OVERLAP=1,BTS_LISTS_AHEAD=1"


def soak(wl, frames, resident, stride):
    K, M, D = pkg.synth.workload_dims(wl)
    st = pkg.synth.make_workload(wl, seed=4321)
    args = dict(max_tracks=K + max(64, K // 8), max_dets=max(K, M), feat_dim=D)
    ref = pkg.DeviceTracker(**args)
    exp = pkg.DeviceTracker(forms=FORMS, **args)
    assert exp.forms()["cvt_overlap"] == "1" and ("FILTER_AHEAD" not in FORMS or exp.forms()["filter_ahead"] == "1")
    exp.inputs_resident(resident)
    bad, checked = [], 0
    CH = 64                                               # frames generated and uploaded per chunk (complete before they are stepped)
    for f0 in range(0, frames, CH):
        host = [st.next() for _ in range(min(CH, frames - f0))]
        dev = [(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()) for fr in host]
        dev2 = [(b, s, x.clone()) for (b, s, x) in dev]      # (the step renormalises rows in place: each tracker its own copy)
        torch.cuda.synchronize()
        for k, (fa, fb) in enumerate(zip(dev, dev2)):
            ref.step(*fa)
            exp.step(*fb)
            if (f0 + k) % stride == stride - 1 or f0 + k == frames - 1:
                da, db = tune._digest(ref), tune._digest(exp)
                checked += 1
                if da != db:
                    bad.append(f0 + k)
    errs = (ref.counts()["errors"], exp.counts()["errors"])
    sa, sb = tune._state_digest(ref), tune._state_digest(exp)
    ref.close(); exp.close()
    ok = not bad and errs == (0, 0) and sa == sb
    print("%-8s resident=%-5s stride %3d: %5d frames, %4d digests compared, mismatches at %s, device errors %s, final state %s -> %s"
          % (wl, resident, stride, frames, checked, bad[:8] or "none", errs, "equal" if sa == sb else "DIFFERENT", "PASS" if ok else "FAIL"))
    return ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    ok = True
    for wl, frames in (("c2", n), ("c3", max(64, n // 8)), ("target", max(48, n // 16))):
        for resident in (True, False):
            for stride in (1, 17):
                ok &= soak(wl, frames if stride > 1 else max(32, frames // 4), resident, stride)
    print(FORMS, "soak:", "PASS (tune.py may offer it: BTS_TUNE_ALLOW_CVT=1)" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
