"""Host side of the C-ABI library under AddressSanitizer + UBSan (`make -C csrc asan`: host code instrumented, gfx950 code
objects as usual -- device-side sanitizers need xnack+, which this pool does not have).  The instrumented library is loaded in a
child interpreter with the ASan runtime preloaded; the child walks the host-only paths every entry point runs before it
touches the GPU (argument checks, configuration, error strings, handle life cycle).  There is no GPU-box variant: ROCm's ASan
runtime intercepts hsa_amd_memory_pool_allocate for the xnack+ device sanitizer and the HIP runtime then fails its first
allocation ("AddressSanitizer: out of memory", measured on the pool's MI355X boxes), so the paths that need a device are
covered by the -m gpu parity tests only."""
import glob
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bot-sort-onnx-tensorrt_amd", "csrc")
SAN_LIB = os.path.join(REPO, "bot-sort-onnx-tensorrt_amd", "libbotsort_hip_asan.so")


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        pytest.skip("no ASan runtime in this image")
    return hits[-1]


def _build():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), "asan"], stdout=subprocess.DEVNULL)
    assert os.path.exists(SAN_LIB)


def _run_child(code, timeout=600):
    env = dict(os.environ)
    env.update(LD_PRELOAD=_asan_runtime(), BTS_LIB_PATH=SAN_LIB, PYTHONPATH=REPO,
               # the interpreter and the HIP runtime are not instrumented: leak reports about them say nothing about csrc/
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:protect_shadow_gap=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=REPO, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "rc=%d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-6000:]
    return p.stdout


HOST_ONLY = r"""
import ctypes, importlib, numpy as np
L = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
lib = L.load()
assert L.LIB_PATH.endswith("libbotsort_hip_asan.so"), L.LIB_PATH
assert b"gfx950" in lib.bts_version()
cfg = L.TrackerConfig()
lib.bts_tracker_default_config(ctypes.byref(cfg))
assert cfg.max_tracks > 0 and cfg.track_high_thresh == 0.40
# every call below must come back with an error code (bad handle / no device), never touch memory it was not given
ctx = ctypes.c_void_p()
rc = lib.bts_context_create(ctypes.byref(ctx), 0)
import torch
if not torch.cuda.is_available():
    assert rc != 0 and len(lib.bts_last_error()) > 0
    trk = ctypes.c_void_p()
    assert lib.bts_tracker_create(None, ctypes.byref(cfg), ctypes.byref(trk)) != 0
    assert lib.bts_tracker_step(None, None, None, None, None, 0, None) != 0
    out = np.zeros(64, np.int32)
    assert lib.bts_tracker_counts(None, out.ctypes.data_as(ctypes.c_void_p), None) != 0
    lib.bts_tracker_destroy(None)
    lib.bts_context_destroy(None)
else:
    assert rc == 0
    lib.bts_context_destroy(ctx)
print("host-only ok")
"""

def test_host_code_under_asan_ubsan_cpu_paths():
    _build()
    assert "host-only ok" in _run_child(HOST_ONLY)


ORACLE_JV = r"""
import ctypes, sys, numpy as np
lib = ctypes.CDLL(sys.argv[1])
lib.bts_oracle_lapjv_extended.restype = ctypes.c_int
lib.bts_oracle_lapjv_extended.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rng = np.random.default_rng(0)
shapes = [(1, 1), (1, 7), (9, 1), (5, 5), (17, 40), (64, 33), (200, 120)]
for n, m in shapes:
    for kind in range(4):
        c = rng.uniform(0.0, 1.0, (n, m))
        if kind == 1: c = np.round(c * 4) / 4            # many exact ties
        if kind == 2: c[:] = 0.25                        # all equal
        if kind == 3: c[rng.random((n, m)) < 0.8] = 1.0  # mostly infeasible
        c = np.ascontiguousarray(c)
        x = np.empty(n, np.int32); y = np.empty(m, np.int32); opt = ctypes.c_double()
        assert lib.bts_oracle_lapjv_extended(n, m, c.ctypes.data, 0.8, x.ctypes.data, y.ctypes.data, ctypes.byref(opt)) == 0
        for i in range(n):
            assert -1 <= x[i] < m and (x[i] < 0 or y[x[i]] == i)
print("oracle jv ok")
"""


def test_oracle_lapjv_under_gcc_asan_ubsan(tmp_path):
    """The checker's C helper (oracle/lapjv_oracle.c: the dense JV every assignment is compared with) compiled with gcc's
    AddressSanitizer + UBSan and driven over ragged / degenerate shapes, tie-heavy and mostly infeasible matrices."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    so = str(tmp_path / "liboracle_lapjv_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-o", so, os.path.join(REPO, "oracle", "lapjv_oracle.c")])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    p = subprocess.run([sys.executable, "-c", ORACLE_JV, so], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "oracle jv ok" in p.stdout, (p.returncode, p.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-3000:]
