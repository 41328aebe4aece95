"""The facade's Python call surface against the reference's: tests/golden/api_surface.json was recorded by
tests/golden/make_api_surface.py from the imported reference (names, kinds, parameter lists, attributes of fresh objects).
A user of the reference must be able to make every call it makes, positionally or by keyword; the facade may ADD optional
parameters and attributes, never drop or reorder one."""
import importlib
import inspect
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("bot-sort-onnx-tensorrt_amd")


def _check_params(owner, fn, ref_params):
    ours = list(inspect.signature(fn).parameters.values())
    names = [p.name for p in ours]
    for k, rp in enumerate(ref_params):
        assert k < len(ours) and names[k] == rp["name"], (owner, "parameter %d is %r in the reference" % (k, rp["name"]), names)
        assert ours[k].kind.name in ("POSITIONAL_OR_KEYWORD", rp["kind"]), (owner, rp["name"], ours[k].kind.name)
        if rp["default"] is not None:            # the reference's defaults are kept
            assert ours[k].default is not inspect._empty and repr(ours[k].default) == rp["default"], (owner, rp["name"], ours[k].default)
    for p in ours[len(ref_params):]:             # extras must be optional
        assert p.default is not inspect._empty or p.kind.name in ("VAR_POSITIONAL", "VAR_KEYWORD"), (owner, "extra required parameter", p.name)


@pytest.mark.parametrize("cname", sorted(REF["classes"]))
def test_class_members(pkg, cname):
    cls = getattr(pkg, cname)
    for name, entry in REF["classes"][cname].items():
        assert hasattr(cls, name), "%s.%s is missing" % (cname, name)
        raw = inspect.getattr_static(cls, name)
        if entry["kind"] == "property":
            assert isinstance(raw, property), (cname, name)
        elif entry["kind"] == "staticmethod":
            assert isinstance(raw, staticmethod), (cname, name)
        elif entry["kind"] == "attribute":
            continue
        if "params" in entry and name != "__repr__":
            _check_params("%s.%s" % (cname, name), getattr(cls, name), entry["params"])


@pytest.mark.parametrize("fname", sorted(REF["functions"]))
def test_module_functions(pkg, fname):
    _check_params(fname, getattr(pkg, fname), REF["functions"][fname])


@pytest.mark.parametrize("cname", sorted(REF["records"]))
def test_detector_box_records(pkg, cname):
    cls = getattr(pkg, cname)
    ref = REF["records"][cname]
    ours = list(inspect.signature(cls).parameters.values())
    assert [p.name for p in ours][:len(ref["params"])] == [p["name"] for p in ref["params"]], cname     # positional order kept
    assert all(p.default is not inspect._empty for p in ours[len(ref["params"]):])
    assert [b.__name__ for b in cls.__mro__[1:] if b.__name__ not in ("ABC", "object")] == ref["bases"]
    obj = cls(*range(len(ref["params"])))                                                                # positional call works
    for k, p in enumerate(ref["params"]):
        assert getattr(obj, p["name"]) == k, (cname, p["name"])


def test_constants(pkg):
    for k, v in REF["TrackState"].items():
        assert getattr(pkg.TrackState, k) == v
    assert {str(k): float(v) for k, v in pkg.KalmanFilter.chi2inv95.items()} == REF["chi2inv95"]


def test_fresh_objects_carry_the_reference_attributes(pkg):
    st = pkg.STrack(np.asarray([1, 2, 3, 4], dtype=np.float32), 0.9, 300, None)
    missing = [a for a in REF["instance_attributes"]["STrack"] if not hasattr(st, a)]
    assert not missing, missing
    assert repr(st) == "OT_0_(0-0)"                                    # D:687-688
    kf_attrs = REF["instance_attributes"]["KalmanFilter"]
    assert all(hasattr(pkg.KalmanFilter, a) or a in pkg.KalmanFilter.__init__.__code__.co_names for a in kf_attrs), kf_attrs
    # BoTSORT needs a device to construct: its attribute names are checked against the class source instead
    src = inspect.getsource(pkg.BoTSORT)
    missing = [a for a in REF["instance_attributes"]["BoTSORT"] if ("self.%s" % a) not in src and not hasattr(pkg.BoTSORT, a)]
    assert not missing, missing
