"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/botsort_hip.h declares, and the Python binding covers all of them.  No compute calls."""
import ctypes
import importlib
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "botsort_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    if not os.path.exists(lib_mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = _declared()
    assert len(names) >= 25
    lib = lib_mod.load()
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert sorted(lib_mod.SIGNATURES) == names, "Python binding and header disagree"
    assert b"gfx950" in lib.bts_version()
    assert lib.bts_status_string(-4) == b"tracker capacity exceeded"


def test_default_config_matches_reference_constants():
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    lib = lib_mod.load()
    cfg = lib_mod.TrackerConfig()
    assert lib.bts_tracker_default_config(ctypes.byref(cfg)) == 0
    # demo_bottrack_onnx_tflite.py:1268-1277, 1571, 1604, 1667, 473
    assert (cfg.track_high_thresh, cfg.track_low_thresh, cfg.new_track_thresh) == (0.40, 0.1, 0.9)
    assert (cfg.match_thresh, cfg.second_match_thresh, cfg.unconfirmed_match_thresh) == (0.8, 0.5, 0.7)
    assert (cfg.proximity_thresh, cfg.appearance_thresh, cfg.dedupe_thresh) == (0.5, 0.25, 0.15)
    assert cfg.track_buffer == 300 and cfg.frame_rate == 30 and cfg.ema_alpha == 0.9
    # extensions are opt-in: the default cascade is the reference's (no motion gate, D:1534-1556 never calls gating_distance)
    assert cfg.motion_gate == 0 and cfg.use_graph == 0 and cfg.record_costs == 0 and cfg.appearance_mode == 0


def test_config_struct_layout_matches_the_header():
    """the ctypes mirror and include/botsort_hip.h declare the same fields in the same order (a field added on one side
    only would silently shift every threshold)."""
    import re
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    hdr = open(os.path.join(REPO, "include", "botsort_hip.h")).read()
    body = re.search(r"typedef struct bts_tracker_config \{(.*?)\} bts_tracker_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = re.findall(r"(int32_t|double)\s+(\w+)\s*;", body)
    want = [(n, {"int32_t": ctypes.c_int32, "double": ctypes.c_double}[t]) for t, n in decl]
    assert want == list(lib_mod.TrackerConfig._fields_)
    assert ctypes.sizeof(lib_mod.TrackerConfig) == 8 * 4 + 10 * 8 + 8 * 4          # (8 + 8 int32 fields, 10 doubles; no padding)
    assert lib_mod.load().bts_tracker_config_size() == ctypes.sizeof(lib_mod.TrackerConfig)


def _integration_stub_cfg():
    """The `_Cfg` class of INTEGRATION.md's ctypes stub, executed as written (the documented stub had fallen
    behind the header and nothing checked it)."""
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"^class _Cfg\(ctypes\.Structure\):.*?(?=^\S)", md, re.S | re.M)
    assert m, "INTEGRATION.md no longer carries the _Cfg stub"
    ns = {"ctypes": ctypes}
    exec(m.group(0), ns)
    return ns["_Cfg"]


def test_integration_md_stub_matches_the_library():
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    stub = _integration_stub_cfg()
    assert ctypes.sizeof(stub) == lib_mod.load().bts_tracker_config_size()
    assert [(n, t) for n, t in stub._fields_] == list(lib_mod.TrackerConfig._fields_)


def test_a_stale_config_struct_is_rejected():
    """create / memory_bytes refuse a config whose struct_size / abi_version is not this library's (host-only entry point)."""
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    lib = lib_mod.load()
    cfg = lib_mod.TrackerConfig()
    assert lib.bts_tracker_default_config(ctypes.byref(cfg)) == 0
    assert cfg.struct_size == ctypes.sizeof(cfg) and cfg.abi_version >= 5
    out = ctypes.c_uint64(0)
    assert lib.bts_tracker_memory_bytes(ctypes.byref(cfg), ctypes.byref(out)) == 0 and out.value > 0
    for field, bad in (("struct_size", cfg.struct_size - 8), ("struct_size", 0), ("abi_version", cfg.abi_version - 1)):
        c2 = lib_mod.TrackerConfig.from_buffer_copy(cfg)
        setattr(c2, field, bad)
        assert lib.bts_tracker_memory_bytes(ctypes.byref(c2), ctypes.byref(out)) == -1      # BTS_EINVAL
        assert b"struct_size" in lib.bts_last_error()


def test_product_path_never_imports_the_oracle():
    pkg_dir = os.path.join(REPO, "bot-sort-onnx-tensorrt_amd")
    for root, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"(import\s+oracle|from\s+oracle|oracle[./]|liboracle|bts_oracle)", src), f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    with pytest.raises(lib_mod.BtsError):
        lib_mod.Context(0)
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    with pytest.raises(pkg.BtsError):
        pkg.iou_distance([[0, 0, 1, 1]], [[0, 0, 1, 1]])


def test_tracker_memory_is_linear_in_the_capacities():
    """bts_tracker_memory_bytes (host-only): nothing in the default mode is O(T * M) except the candidate bitmap -- the
    round-1 layout allocated 5.9 GB for 16384 x 4096 x 2048 (ADVICE), DESIGN.md section 2 quotes this function's figure."""
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    f = pkg.tracker_memory_bytes
    big = f(16384, 4096, 2048)
    assert 400e6 < big < 480e6
    feature_tables = 2 * 16384 * 2048 * 4 + 16384 * 2048
    assert big > feature_tables and big - feature_tables < 150e6            # everything else: < 150 MB
    assert f(16384, 4096, 2048, appearance_mode=1) - big > 16384 * 4096 * 4 * 0.7   # DENSE mode: + the similarity matrix (- the int8 shadows)
    assert f(4096, 1024, 2048, record_costs=True) - f(4096, 1024, 2048) > 3 * 4096 * 1024 * 8
    assert f(16384, 4096, 2048, max_edges=100000) < big
    with pytest.raises(pkg.BtsError):
        f(0, 10, 128)
    # a stage threshold >= 1 makes EVERY pair feasible: the edge lists then keep the T * M that cannot overflow
    lean, full = f(8192, 2048, 128), f(8192, 2048, 128, match_thresh=1.0)
    per_edge = 4 * 4 + 2 * 8 + 4      # e_row, e_col, csr_col, csr_lcol, hard + e_cost, csr_cost per list entry (lap.hip layout) ...
    assert full - lean > 2 * (8192 * 2048 - 64 * (8192 + 2048)) * (per_edge - 8)      # ... for both work areas
    assert f(8192, 2048, 128, second_match_thresh=1.5) == full and f(8192, 2048, 128, unconfirmed_match_thresh=1.0) == full
    # the int8 shadows are addressed with 32-bit byte offsets: tables beyond 2^32 bytes are refused in FILTER mode, not mis-read
    with pytest.raises(pkg.BtsError):
        f(4 * 1024 * 1024, 256, 2048)
    assert f(4 * 1024 * 1024, 256, 2048, appearance_mode=1) > 0
    # tie_mode 1 (exact-tie step): its scratch is linear in T + M + edges, and its configuration rules are checked on the host
    assert 0 < f(10000, 2000, 2048, tie_mode=1) - f(10000, 2000, 2048) < 40e6
    assert f(1024, 512, 256, face_dim=128, tie_mode=1) > f(1024, 512, 256, face_dim=128)      # faces are carried too (overrides)
    for bad in (dict(motion_gate=1), dict(appearance_thresh=0.9), dict(match_thresh=-0.1)):
        with pytest.raises(pkg.BtsError):
            f(1024, 512, 256, tie_mode=1, **bad)
    with pytest.raises(pkg.BtsError):
        f(40000, 8000, 128, tie_mode=1)                 # (N + M)^2 must fit int32 (the dense JV's sweep counter)
    with pytest.raises(pkg.BtsError):
        f(1024, 512, 256, tie_mode=2)


def test_binding_argument_counts_match_the_header():
    """Every prototype of include/botsort_hip.h against the ctypes signature table: same number of parameters, pointer
    parameters bound as pointers (a drifted binding would pass garbage through the C ABI without any error)."""
    import ctypes
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    text = open(os.path.join(REPO, "include", "botsort_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"\b(?:int|const char \*|void)\s*\*?\s*(bts_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    seen = {}
    for name, params in protos:
        params = " ".join(params.split())
        plist = [] if params in ("", "void") else [q.strip() for q in params.split(",")]
        seen[name] = plist
    assert set(seen) == set(lib_mod.SIGNATURES), set(seen) ^ set(lib_mod.SIGNATURES)
    pointer_types = (ctypes.c_void_p, ctypes.c_char_p)
    for name, plist in seen.items():
        _, argtypes = lib_mod.SIGNATURES[name]
        assert len(argtypes) == len(plist), (name, plist, argtypes)
        for decl, at in zip(plist, argtypes):
            is_ptr_c = "*" in decl or decl.startswith("bts_stream ")      # (typedef void *bts_stream)
            is_ptr_py = at in pointer_types or hasattr(at, "contents") or at is lib_mod._P
            assert is_ptr_c == is_ptr_py, (name, decl, at)


def test_memory_bytes_follow_the_execution_forms():
    """bts_tracker_memory_bytes_forms (host-only): the forms that carry extra arrays are accounted for -- the second set of
    detection shadows of the side-stream conversion, the per-helper slices of the self-preparing solver, the slot -> list-index
    maps of the early merge; forms that only change launches or layouts cost nothing."""
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    f = pkg.tracker_memory_bytes
    T, M, D = 11250, 10000, 2048
    base = f(T, M, D)
    assert f(T, M, D, forms="") == base and f(T, M, D, forms="BTS_LISTS_AHEAD=1,BTS_PREDICT_FRONT=1,BTS_FILTER_GUESS=1,BTS_SHADOW_PLANES=1,BTS_RENORM_FUSED=1") == base
    cvt = f(T, M, D, forms="BTS_CVT_OVERLAP=1") - base
    assert M * D <= cvt <= M * D + 2 * M * 80 + 65536                 # one more [M][D] int8 image + the records (body, face)
    merge = f(T, M, D, forms={"BTS_MERGE_EARLY": "1"}) - base
    assert 2 * 4 * T - 1024 <= merge <= 2 * 4 * T + 1024              # two int32 maps over the slots
    self63 = f(T, M, D, forms="BTS_LAP_SELF=1") - base
    self15 = f(T, M, D, forms="BTS_LAP_SELF=1,BTS_LAP_HELPERS=15") - base
    assert 0 < self15 < self63 < 64 * 2 * 400 * 1024                    # (two work areas, <= ~300 KB per helper and area)
    assert abs(self63 / self15 - 63 / 15) < 0.25


def test_boolean_forms_read_zero_false_off_as_off(monkeypatch):
    """"KEY=0" / "false" / "off" / a dict value of 0 / False / None switch a boolean form OFF (they used to
    switch it ON: any non-empty value did), and so does "KEY=" against an environment that says 1.  Witness: the forms whose
    extra arrays show in bts_tracker_memory_bytes_forms (host-only)."""
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    f = pkg.tracker_memory_bytes
    T, M, D = 4608, 4096, 2048
    base = f(T, M, D)
    on = f(T, M, D, forms="BTS_CVT_OVERLAP=1")
    assert on > base
    for off in ("BTS_CVT_OVERLAP=0", "BTS_CVT_OVERLAP=false", "BTS_CVT_OVERLAP=OFF", "BTS_CVT_OVERLAP=no", "BTS_CVT_OVERLAP=",
                {"BTS_CVT_OVERLAP": 0}, {"BTS_CVT_OVERLAP": False}, {"BTS_CVT_OVERLAP": None}, {"BTS_MERGE_EARLY": 0, "BTS_LAP_SELF": False}):
        assert f(T, M, D, forms=off) == base, off
    for still_on in ("BTS_CVT_OVERLAP", "BTS_CVT_OVERLAP=true", "BTS_CVT_OVERLAP=2", {"BTS_CVT_OVERLAP": True}, {"BTS_CVT_OVERLAP": 1}):
        assert f(T, M, D, forms=still_on) == on, still_on
    monkeypatch.setenv("BTS_CVT_OVERLAP", "0")
    assert f(T, M, D) == base
    monkeypatch.setenv("BTS_CVT_OVERLAP", "1")
    assert f(T, M, D) == on and f(T, M, D, forms={"BTS_CVT_OVERLAP": 0}) == base
    assert pkg.tracker.forms_string({"A": 0, "B": True, "C": "row", "D": None, "E": False, "F": ""}) == "A=0,B=1,C=row,D=,E=,F="
