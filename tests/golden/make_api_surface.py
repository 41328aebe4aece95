"""Record the reference's Python call surface for the tracking path as DATA: for each class / function the
parameter names, defaults and kinds that `inspect.signature` reports on the imported reference, plus the public attributes
a freshly constructed object carries.  Build container only (imports /root/reference through ref_harness); the output,
tests/golden/api_surface.json, is what tests/test_api_surface.py compares the facade with."""
import inspect
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import ref_harness  # noqa: E402


def sig(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        d = None if p.default is inspect._empty else repr(p.default)
        out.append({"name": p.name, "kind": p.kind.name, "default": d})
    return out


def members(cls, names):
    res = {}
    for n in names:
        raw = inspect.getattr_static(cls, n)
        kind = "staticmethod" if isinstance(raw, staticmethod) else "classmethod" if isinstance(raw, classmethod) else \
               "property" if isinstance(raw, property) else "method" if callable(raw) else "attribute"
        entry = {"kind": kind}
        if kind in ("staticmethod", "classmethod", "method"):
            entry["params"] = sig(getattr(cls, n))
        res[n] = entry
    return res


def main():
    mod = ref_harness.load_reference()
    out = {"classes": {}, "functions": {}, "instance_attributes": {}}
    wanted = {
        "KalmanFilter": ["__init__", "initiate", "predict", "project", "multi_predict", "update", "gating_distance"],
        "BaseTrack": ["next_id", "activate", "predict", "update", "mark_lost", "mark_long_lost", "mark_removed", "end_frame"],
        "STrack": ["__init__", "update_body_features", "update_face_features", "predict", "multi_predict", "multi_gmc", "activate",
                   "re_activate", "update", "tlwh", "tlbr", "xywh", "tlwh_to_xyah", "tlwh_to_xywh", "to_xywh", "tlbr_to_tlwh",
                   "tlwh_to_tlbr", "__repr__"],
        "BoTSORT": ["__init__", "update"],
    }
    for cname, names in wanted.items():
        cls = getattr(mod, cname)
        out["classes"][cname] = members(cls, [n for n in names if hasattr(cls, n)])
    for f in ("linear_assignment", "iou_distance", "bbox_ious", "bbox_iou", "joint_stracks", "sub_stracks", "remove_duplicate_stracks"):
        out["functions"][f] = sig(getattr(mod, f))
    out["records"] = {}        # the detector's box records (D:84-116): constructor parameters and base classes
    for cname in ("Box", "Body", "Head", "Face", "Hand"):
        cls = getattr(mod, cname)
        out["records"][cname] = {"params": sig(cls)[0:], "bases": [b.__name__ for b in cls.__mro__[1:] if b.__name__ not in ("ABC", "object")]}
    for f in ("find_most_relevant_object", "bbox_iou_by_box"):
        if hasattr(mod, f):
            out["functions"][f] = sig(getattr(mod, f))
    out["TrackState"] = {k: int(v) for k, v in vars(mod.TrackState).items() if not k.startswith("_") and isinstance(v, int)}
    out["chi2inv95"] = {str(k): float(v) for k, v in mod.KalmanFilter.chi2inv95.items()} if hasattr(mod.KalmanFilter, "chi2inv95") else \
        {str(k): float(v) for k, v in mod.chi2inv95.items()}
    kf = mod.KalmanFilter()
    out["instance_attributes"]["KalmanFilter"] = sorted(k for k in vars(kf) if not k.startswith("__"))
    st = mod.STrack(np.asarray([1, 2, 3, 4], dtype=np.float32), 0.9, 300, None)
    out["instance_attributes"]["STrack"] = sorted(set(k for k in vars(st)) | set(k for k in vars(mod.BaseTrack) if not k.startswith("__") and not callable(getattr(mod.BaseTrack, k)) and not isinstance(inspect.getattr_static(mod.BaseTrack, k), property)))
    bot = mod.BoTSORT(None, None, None, frame_rate=30)
    out["instance_attributes"]["BoTSORT"] = sorted(k for k in vars(bot))
    path = os.path.join(HERE, "api_surface.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out["classes"].items()}, len(out["functions"]))


if __name__ == "__main__":
    main()
