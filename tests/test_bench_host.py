"""Host-side logic of bench.py that needs no GPU: the byte models behind `roofline.step`, the table sizing, the defaults the
driver's plain command relies on, and that a box without a GPU gets a loud failure -- never a JSON line."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_defaults_are_the_headline_configuration(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.workload, a.gpus, a.streams_per_gpu) == ("target", 1, 1)
    assert not a.motion_gate and not a.dense and not a.graph
    pkg = __import__("importlib").import_module("bot-sort-onnx-tensorrt_amd")
    assert pkg.synth.workload_dims("target") == (10000, 2000, 2048)


def test_step_byte_models():
    b = _bench()
    survey, built = b.step_bytes(10000, 2000, 2048)
    # SURVEY.md 8(d) at the headline size: the 2 N M s_c matrix term (160 MB) dominates; as built there is no matrix
    N, M, D = 10000, 2000, 2048
    assert survey == 2 * N * 72 * 4 + M * 148 * 4 + (N + M) * 16 + (N + M) * D * 4 + 2 * N * M * 4 + 4 * M * D * 4 + 4 * (N + M)
    assert abs(survey - 331.0e6) < 0.5e6 and abs(built - 161.6e6) < 0.5e6
    assert built < survey
    s2, b2 = b.step_bytes(512, 256, 2048)
    assert s2 < survey and b2 < built


def test_table_capacity_covers_the_gate_runs():
    b = _bench()
    assert b.table_rows(10000) == 11250 and b.table_rows(512) == 576
    assert b.table_rows(4096, motion_gate=True) == 6144


def test_no_gpu_means_no_number():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    for extra in ([], ["--gpus", "2"]):
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu"] + extra,
                           capture_output=True, text=True, timeout=300, cwd=REPO)
        assert p.returncode != 0, p.stdout[-500:]
        for ln in p.stdout.splitlines():
            if ln.startswith("{"):
                raise AssertionError("bench.py printed a result line without a GPU: " + json.dumps(ln)[:200])


def test_line_guard_prints_what_was_measured_if_a_later_leg_wedges(tmp_path):
    """bench.LineGuard: once the first measurement exists its contract fields are armed; if the legs behind it
    never finish the watchdog prints them with a `truncated` note and ends the process with rc 0; a normal run prints the complete
    line instead and the watchdog stays silent.  Exactly ONE line on stdout either way."""
    import json
    import subprocess
    import sys
    code = r'''
import sys, time
sys.path.insert(0, %r)
import bench
g = bench.LineGuard(0)
g.arm({"metric": "m", "value": 123.0, "roofline": None}, 1.0)
if sys.argv[1] == "wedge":
    time.sleep(30)                      # a leg that never comes back
else:
    g.emit({"metric": "m", "value": 456.0, "roofline": {"frac": 0.5}})
    time.sleep(2.5)                     # (the cancelled watchdog must not print a second line)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, "wedge"], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout, r.stderr[-500:])
    d = json.loads(lines[0])
    assert d["value"] == 123.0 and "truncated" in d and "watchdog" in d["truncated"]
    r = subprocess.run([sys.executable, "-c", code, "ok"], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout, r.stderr[-500:])
    assert json.loads(lines[0])["value"] == 456.0 and "truncated" not in lines[0]
