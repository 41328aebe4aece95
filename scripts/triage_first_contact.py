#!/usr/bin/env python
"""Triage leg of the first GPU contact (scripts/gpu_r6_a.sh): a few frames of a small stream through the resident tracker, each
frame compared with the CPU oracle stage by stage, with the tracker's sticky error word / message and the solver's and
pre-filter's own counters printed per frame -- so that ONE lease on which smoke or the bench failed still yields a diagnosis
(which stage, which kernel family) instead of just "rc != 0".  Checker only: nothing here is timed.

    python scripts/triage_first_contact.py [c2|c1|tiny] [frames]"""
import importlib
import os
import sys
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SHAPES = {"tiny": (64, 32, 256, 512, 64), "c1": (64, 32, 2048, 512, 64), "c2": (512, 256, 2048, 1024, 512)}


def main():
    import torch
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    nt, nd, dim, cap_t, cap_d = SHAPES[which]
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    from oracle import bts_oracle as bo
    print("triage %s: %d tracks x %d dets x %d-d, %d frames; library %s" % (which, nt, nd, dim, frames, pkg._lib.load(require_gpu=True).bts_version().decode()))
    st = pkg.synth.SyntheticStream(nt, nd, dim, seed=1234, churn=0.05)
    trk = pkg.DeviceTracker(max_tracks=cap_t, max_dets=cap_d, feat_dim=dim)
    print("forms in effect:", trk.forms())
    ora = bo.TrackerOracle()
    bad = 0
    for f in range(frames):
        fr = st.next()
        try:
            trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())
            torch.cuda.synchronize()
        except Exception:                                      # noqa: BLE001
            print("frame %d: step raised" % f)
            traceback.print_exc()
            bad += 1
            break
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy())
        line = []
        for k in (1, 2, 3):
            try:
                r = trk.stage_result(k)
                want = ora.trace["m%d" % k]
                same = np.array_equal(r["m"], want)
                line.append("stage%d %s (%d pairs, oracle %d)" % (k, "ok" if same else "DIFFERS", len(r["m"]), len(want)))
                if not same:
                    bad += 1
                    got = {tuple(p) for p in np.asarray(r["m"]).reshape(-1, 2).tolist()}
                    exp = {tuple(p) for p in np.asarray(want).reshape(-1, 2).tolist()}
                    print("   stage %d: only device %s | only oracle %s" % (k, sorted(got - exp)[:8], sorted(exp - got)[:8]))
            except Exception as e:                             # noqa: BLE001
                line.append("stage%d raised %r" % (k, e))
                bad += 1
        c = trk.counts()
        print("frame %d: %s | counts %s" % (f, "; ".join(line), c))
        for name, fn in (("debug_filter", trk.debug_filter), ("debug_lap(1)", lambda: trk.debug_lap(1).tolist()),
                         ("debug_lap(2)", lambda: trk.debug_lap(2).tolist()), ("debug_lap(3)", lambda: trk.debug_lap(3).tolist())):
            try:
                print("   %s: %s" % (name, fn()))
            except Exception as e:                             # noqa: BLE001
                print("   %s raised %r" % (name, e))
        for which_list, lst in ((0, ora.tracked), (1, ora.lost)):
            try:
                t = trk.export(which_list)
                want = bo.TrackerOracle.table(lst)
                for key in ("track_id", "state", "frame_id", "tracklet_len"):
                    if not np.array_equal(t[key], want[key]):
                        print("   list %d: %s DIFFERS (device %d rows, oracle %d)" % (which_list, key, len(t[key]), len(want[key])))
                        bad += 1
                        break
            except Exception as e:                             # noqa: BLE001
                print("   export(%d) raised %r" % (which_list, e))
                bad += 1
    print("triage %s: %s" % (which, "CLEAN" if bad == 0 else "%d problems" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
