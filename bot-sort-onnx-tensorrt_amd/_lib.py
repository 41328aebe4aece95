"""ctypes binding of libbotsort_hip.so (C ABI declared in include/botsort_hip.h).

The library is the product: nothing here falls back to NumPy.  If the shared object is missing or
a call fails, a `BtsError` is raised -- loudly.  PyTorch-ROCm is imported first on purpose: the
library then binds to the HIP runtime torch already loaded, so torch tensors (used only as device
buffers: `.data_ptr()`) and torch streams are valid in its calls.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BTS_LIB_PATH") or os.path.join(_HERE, "libbotsort_hip.so")      # (override: A/B runs of two builds on one box)


class BtsError(RuntimeError):
    pass


class TrackerConfig(ctypes.Structure):
    """bts_tracker_config (defaults = the reference's hard-coded values, D:1268-1277)."""
    _fields_ = [
        ("struct_size", c_int32), ("abi_version", c_int32), ("max_tracks", c_int32), ("max_dets", c_int32), ("feat_dim", c_int32), ("face_dim", c_int32),
        ("frame_rate", c_int32), ("track_buffer", c_int32),
        ("track_high_thresh", c_double), ("track_low_thresh", c_double), ("new_track_thresh", c_double),
        ("match_thresh", c_double), ("second_match_thresh", c_double), ("unconfirmed_match_thresh", c_double),
        ("proximity_thresh", c_double), ("appearance_thresh", c_double), ("dedupe_thresh", c_double),
        ("ema_alpha", c_double), ("record_costs", c_int32), ("appearance_mode", c_int32), ("use_graph", c_int32),
        ("motion_gate", c_int32), ("max_edges", c_int32), ("tie_mode", c_int32),
        ("kalman_dtype", c_int32), ("cost_dtype", c_int32),
    ]


# name -> (restype, argtypes); every symbol include/botsort_hip.h declares
_P = c_void_p
SIGNATURES = {
    "bts_version": (c_char_p, []),
    "bts_last_error": (c_char_p, []),
    "bts_status_string": (c_char_p, [c_int]),
    "bts_context_create": (c_int, [POINTER(c_void_p), c_int]),
    "bts_context_destroy": (c_int, [_P]),
    "bts_context_set_cos_kernel": (c_int, [_P, c_int]),
    "bts_kf_initiate": (c_int, [_P, _P, c_int, _P, _P, _P]),
    "bts_kf_multi_predict": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P]),
    "bts_kf_project": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P]),
    "bts_kf_update": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "bts_kf_multi_gmc": (c_int, [_P, _P, _P, c_int, _P, _P]),
    "bts_kf_gating": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "bts_iou_distance": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P]),
    "bts_cos_similarity": (c_int, [_P, _P, c_int, _P, c_int, c_int, _P, c_int, _P]),
    "bts_fuse_cost_stage1": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_double, c_double, _P, c_int, _P]),
    "bts_fuse_cost_stage3": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_double, c_double, _P, c_int, _P]),
    "bts_lapjv": (c_int, [_P, _P, c_int, c_int, c_int, c_double, _P, _P, _P]),
    "bts_lapjv_exact": (c_int, [_P, _P, c_int, c_int, c_int, c_double, _P, _P, c_int, _P, _P]),
    "bts_feature_ema": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, _P]),
    "bts_feature_renorm": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "bts_debug_i8_rows": (c_int, [_P, _P, c_int, c_int, c_double, _P, _P, _P]),
    "bts_debug_i8_pairs": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "bts_debug_i8_layout": (c_int, [c_int, _P]),
    "bts_tracker_debug_filter": (c_int, [_P, _P, _P]),
    "bts_debug_jobtrace": (c_int, [_P]),
    "bts_debug_launch_floor": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "bts_associate_parts": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P]),
    "bts_dedupe_pairs": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_double, _P, _P, _P]),
    "bts_tracker_config_size": (c_int, []),
    "bts_tracker_default_config": (c_int, [POINTER(TrackerConfig)]),
    "bts_tracker_memory_bytes": (c_int, [POINTER(TrackerConfig), POINTER(ctypes.c_uint64)]),
    "bts_tracker_create": (c_int, [_P, POINTER(TrackerConfig), POINTER(c_void_p)]),
    "bts_tracker_create_forms": (c_int, [_P, POINTER(TrackerConfig), c_char_p, POINTER(c_void_p)]),
    "bts_tracker_memory_bytes_forms": (c_int, [POINTER(TrackerConfig), c_char_p, POINTER(ctypes.c_uint64)]),
    "bts_tracker_forms": (c_int, [_P, c_char_p, c_int]),
    "bts_tracker_destroy": (c_int, [_P]),
    "bts_tracker_reset": (c_int, [_P, _P]),
    "bts_tracker_step": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "bts_tracker_inputs_resident": (c_int, [_P, c_int]),
    "bts_tracker_counts": (c_int, [_P, _P, _P]),
    "bts_tracker_export": (c_int, [_P, c_int, _P, _P, _P, _P, _P]),
    "bts_tracker_export_lists": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "bts_tracker_read_column": (c_int, [_P, c_int, _P, _P]),
    "bts_tracker_export_mot": (c_int, [_P, _P, _P]),
    "bts_tracker_export_mot_async": (c_int, [_P, _P, _P, _P]),
    "bts_tracker_export_state": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P]),
    "bts_tracker_read_slot": (c_int, [_P, c_int, c_int, _P, _P]),
    "bts_pinned_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "bts_pinned_free": (c_int, [_P]),
    "bts_tracker_gmc": (c_int, [_P, _P, _P]),
    "bts_tracker_debug_lap": (c_int, [_P, c_int, _P, _P]),
    "bts_tracker_profile_enable": (c_int, [_P, c_int, c_int]),
    "bts_tracker_profile_stride": (c_int, [_P, c_int]),
    "bts_tracker_profile_read": (c_int, [_P, POINTER(c_double), POINTER(c_int32)]),
    "bts_tracker_stage_result": (c_int, [_P, c_int, _P, _P, _P, POINTER(c_int32), POINTER(c_int32), _P]),
}

_lib = None


def load(require_gpu: bool = False) -> ctypes.CDLL:
    """Load the shared object and bind every prototype.  No device is touched here."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BtsError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C bot-sort-onnx-tensorrt_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        import torch  # noqa: F401  (loads the HIP runtime the library binds to)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(lib, name) and (name.startswith("bts_debug_") or os.environ.get("BTS_LIB_PATH")):
                continue                     # (A/B runs of OLDER builds through BTS_LIB_PATH lack the newer entry points; a call to
                                             #  one of them then fails loudly with AttributeError.  The package's own library must
                                             #  export every symbol of the header: tests/test_abi.py)
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if hasattr(lib, "bts_tracker_config_size") and lib.bts_tracker_config_size() != ctypes.sizeof(TrackerConfig):
            raise BtsError("%s: bts_tracker_config is %d bytes in the library, %d in this binding (stale _lib.TrackerConfig?)"
                           % (LIB_PATH, lib.bts_tracker_config_size(), ctypes.sizeof(TrackerConfig)))
        _lib = lib
    if require_gpu:
        import torch
        if not torch.cuda.is_available():
            raise BtsError("no MI355X visible: the BoT-SORT hot path has no CPU fallback")
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        lib = load()
        raise BtsError("%s: %s" % (lib.bts_status_string(rc).decode(), lib.bts_last_error().decode()))
    return rc


def ptr(t) -> int:
    """data_ptr of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "device buffers handed to the C ABI must be contiguous"
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class Context:
    """bts_context: one per (GPU, host thread)."""

    def __init__(self, device: int = 0):
        lib = load(require_gpu=True)
        h = c_void_p()
        check(lib.bts_context_create(ctypes.byref(h), int(device)))
        self.handle = h
        self.device = int(device)

    def set_cos_kernel(self, which: int):
        check(load().bts_context_set_cos_kernel(self.handle, int(which)))

    def close(self):
        if getattr(self, "handle", None):
            load().bts_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = None) -> Context:
    """The context of `device` (default: torch's current device -- a rank with LOCAL_RANK != 0 must not land on GPU 0)."""
    if device is None:
        import torch
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    device = int(device)
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
