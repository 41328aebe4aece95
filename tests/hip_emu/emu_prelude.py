"""TEST INFRASTRUCTURE: makes a CPU-only interpreter look like it has one GPU, for running the package against the CPU
emulation of its kernels (tests/hip_emu/libbotsort_emu.so, selected through BTS_LIB_PATH).  "Device" memory of the emulated
library is host memory, so torch CPU tensors stand in for CUDA tensors: `.cuda()`, `device="cuda"` and the stream / device
queries are patched to no-ops.  Imported (as a pytest plugin: -p tests.hip_emu.emu_prelude) only by the child processes of
tests/test_emu_*.py -- never by the product."""
import os

import torch

assert os.environ.get("BTS_LIB_PATH", "").endswith("libbotsort_emu.so"), "the emulation prelude is for the emulated library only"


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def elapsed_time(self, other):        # (the emulator does not tell time; callers only need a positive number)
        return 1.0


torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.current_device = lambda: 0
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.current_stream = lambda *a, **k: _Stream()
torch.cuda.default_stream = lambda *a, **k: _Stream()
torch.cuda.Stream = lambda *a, **k: _Stream()
torch.cuda.set_stream = lambda s: None
torch.cuda.stream = lambda s: _Stream()
torch.cuda.Event = _Event
torch.Tensor.cuda = lambda self, *a, **k: self.clone()       # (a host -> device copy is a COPY: kernels must not write into the caller's array)
_to = torch.Tensor.to


def _to_cpu(self, *a, **k):
    k.pop("device", None)
    a = tuple(x for x in a if not (isinstance(x, (str, torch.device)) and "cuda" in str(x)))
    had_device = True
    return _to(self, *a, **k).clone() if (a or k) else self.clone()


torch.Tensor.to = _to_cpu
for _name in ("empty", "zeros", "ones", "full", "tensor", "as_tensor"):
    _orig = getattr(torch, _name)

    def _wrap(*a, _o=_orig, **k):
        if "cuda" in str(k.get("device", "")):
            k.pop("device")
        return _o(*a, **k)
    setattr(torch, _name, _wrap)


# ---- HIPEMU_STREAMS=sidelate | mainlate (hip/hip_runtime.h): one side of the library's two-stream forms lags as far behind as the
# program's own event dependencies allow.  What the device would order for a real torch program is re-created here: a host
# synchronisation (torch.cuda.synchronize, Tensor.cpu) is a "device idle" point, and the tensors handed to a step stay alive
# while the step may still be queued (on a GPU the caching allocator's stream semantics do that).
if os.environ.get("HIPEMU_STREAMS"):
    import collections
    import ctypes as _ct

    _emu = _ct.CDLL(os.environ["BTS_LIB_PATH"])
    torch.cuda.synchronize = lambda *a, **k: _emu.hipemu_drain_all()
    _Stream.synchronize = lambda self: _emu.hipemu_drain_all()
    _Event.synchronize = lambda self: _emu.hipemu_drain_all()
    _cpu = torch.Tensor.cpu

    def _cpu_sync(self, *a, **k):
        _emu.hipemu_drain_all()
        return _cpu(self, *a, **k)
    torch.Tensor.cpu = _cpu_sync
    _alive = collections.deque(maxlen=64)

    def _keep_step_inputs():
        import importlib
        trk_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd.tracker")
        if getattr(trk_mod.DeviceTracker, "_emu_keeps_inputs", False):
            return
        _step = trk_mod.DeviceTracker.step

        def step(self, *a, **k):
            _alive.append((a, k))
            return _step(self, *a, **k)
        trk_mod.DeviceTracker.step = step
        trk_mod.DeviceTracker._emu_keeps_inputs = True
    _keep_step_inputs()


# ---- BTS_HOST_TIME_REPORT=path: per test, the wall time spent OUTSIDE the library's entry points (oracle, stream generation,
# NumPy, pytest) -- on a GPU box the library calls take milliseconds, so this is what a test costs there; the driver's limit for
# the whole `-m gpu` suite is 1200 s.  Written as JSON at session end (one file per xdist worker).
if os.environ.get("BTS_HOST_TIME_REPORT"):
    import json
    import time

    import pytest

    _acc = {"lib": 0.0, "depth": 0}
    _rows = {}

    def _wrap(fn):
        def call(*a, **k):
            if _acc["depth"]:
                return fn(*a, **k)
            _acc["depth"] += 1
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                _acc["lib"] += time.perf_counter() - t0
                _acc["depth"] -= 1
        return call

    def _wrap_library():
        import importlib
        lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
        lib = lib_mod.load()
        if getattr(lib, "_host_time_wrapped", False):
            return
        for name in lib_mod.SIGNATURES:
            if hasattr(lib, name):
                setattr(lib, name, _wrap(getattr(lib, name)))
        lib._host_time_wrapped = True

    @pytest.hookimpl(hookwrapper=True)
    def pytest_runtest_protocol(item, nextitem):
        _wrap_library()
        lib0, t0 = _acc["lib"], time.perf_counter()
        yield
        wall = time.perf_counter() - t0
        _rows[item.nodeid] = {"wall_s": round(wall, 3), "library_s": round(_acc["lib"] - lib0, 3), "host_s": round(wall - (_acc["lib"] - lib0), 3)}

    def pytest_sessionfinish(session, exitstatus):
        path = os.environ["BTS_HOST_TIME_REPORT"]
        worker = os.environ.get("PYTEST_XDIST_WORKER", "")
        if _rows:
            json.dump(_rows, open(path + ("." + worker if worker else ""), "w"), indent=0)
