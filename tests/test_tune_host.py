"""Execution-form selection (bot-sort-onnx-tensorrt_amd/tune.py, bts_tracker_create_forms) on the CPU: the selection logic
against a scripted tracker double, the C ABI's form lookup on the emulated library, and the isolated runner -- child process,
progress lines, a candidate that kills the child is blamed and skipped on the restart -- end to end on the CPU emulation of the
kernel sources (tests/hip_emu; it tells no time, so only the flow, the identity verdicts and the crash handling are asserted)."""
import importlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(REPO, "tests", "hip_emu")
tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")


class _Double:
    """A tracker double: the time a step takes and the result it produces are scripted per form."""
    clock = [0.0]

    def __init__(self, forms, cost_us, wrong=(), errors=(), late_wrong=()):
        self.f = forms or ""
        self.cost = cost_us(self.f)
        self.wrong = any(w in self.f for w in wrong)
        self.late = any(w in self.f for w in late_wrong)      # diverges only after 40 steps: invisible to a few verified frames
        self.err = any(e in self.f for e in errors)
        self.n = 0

    def step(self, *fr):
        self.n += 1
        _Double.clock[0] += self.cost * 1e-6

    def stage_result(self, k):
        v = np.array([self.n, k, 7 if (self.wrong and self.n > 2) or (self.late and self.n > 40) else 0], dtype=np.int32)
        return {"x": v, "y": v}

    def export(self, which):
        z = np.zeros(3)
        return {k: z for k in ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len", "det_index", "mean", "tlbr", "score")}

    def counts(self):
        return {"errors": 1 if self.err else 0}

    def forms(self):
        return {"asked": self.f}

    def close(self):
        pass


def _select(cost_us, monkeypatch, **kw):
    monkeypatch.setattr(tune.time, "perf_counter", lambda: _Double.clock[0])
    if "BTS_TUNE_SOAK_PASSES" not in os.environ:            # (the selection's own soak of the side-stream forms: its test switches it on)
        monkeypatch.setenv("BTS_TUNE_SOAK_PASSES", "0")
    frames = [(None, None, None)] * 12
    return tune.select_forms(lambda forms: _Double(forms, cost_us, **kw), frames, warm=4, verify=3, repeats=1,
                             clone=lambda fr: fr, sync=lambda: None)


def test_selection_keeps_what_is_identical_and_faster(monkeypatch):
    def cost(f):       # lists_ahead -10 %, sorted edges -20 %, row solver +5 %, together they add up
        c = 100.0
        c *= 0.9 if "BTS_LISTS_AHEAD" in f else 1.0
        c *= 1.02 if "BTS_PREDICT_FRONT" in f else 1.0       # (alternatives of the same group that are a little slower)
        c *= 1.01 if "BTS_CVT_OVERLAP" in f else 1.0
        c *= 0.8 if "BTS_EDGES_SORTED" in f else 1.0
        c *= 1.05 if "BTS_LAP_FORM=row" in f else 1.0
        return c
    r = _select(cost, monkeypatch)
    assert r["forms"] == "BTS_EDGES_SORTED=1,BTS_LISTS_AHEAD=1" and r["chosen"] == "edges_sorted+lists_ahead"
    assert abs(r["default_us"] - 100.0) < 1e-6 and abs(r["chosen_us"] - 72.0) < 1e-6
    by = {m["name"]: m for m in r["table"]}
    assert by["lap_row"]["eligible"] and by["lap_row"]["speedup"] < 1.0          # identical, just slower: not taken
    assert all(m.get("identical", True) for m in r["table"])


def test_a_faster_candidate_with_another_result_is_never_taken(monkeypatch):
    def cost(f):
        return 50.0 if "BTS_EDGES_SORTED" in f else 95.0 if "BTS_PREDICT_FIRST" in f else 100.0
    r = _select(cost, monkeypatch, wrong=("BTS_EDGES_SORTED",))
    by = {m["name"]: m for m in r["table"]}
    assert by["edges_sorted"]["identical"] is False and by["edges_sorted"]["eligible"] is False
    assert r["forms"] == "BTS_PREDICT_FIRST=1"
    # ... nor one that raised a device error
    r = _select(cost, monkeypatch, errors=("BTS_EDGES_SORTED",))
    assert r["forms"] == "BTS_PREDICT_FIRST=1" and {m["name"]: m for m in r["table"]}["edges_sorted"]["errors"] == 1


def test_a_combination_that_is_slower_than_its_best_member_is_dropped(monkeypatch):
    def cost(f):       # each helps alone, together they hurt
        a, b = "BTS_LISTS_AHEAD" in f, "BTS_CVT_OVERLAP" in f
        return 100.0 if not (a or b) else 120.0 if (a and b) else 90.0 if a else 94.0
    monkeypatch.setenv("BTS_TUNE_ALLOW_CVT", "1")          # (the side-stream conversion is a candidate only after its soak)
    r = _select(cost, monkeypatch)
    assert r["forms"] == "BTS_LISTS_AHEAD=1" and abs(r["chosen_us"] - 90.0) < 1e-6
    assert any("+" in m["name"] for m in r["table"])                              # (the combination was measured)


def test_cvt_overlap_is_not_offered_before_its_soak(monkeypatch):
    """the side-stream forms (BTS_CVT_OVERLAP, round 6: BTS_FILTER_AHEAD) have no evidence until a long digest
    soak has run on a device.  The selection offers them only if the caller attests one (BTS_TUNE_ALLOW_CVT=1) or -- round 6 -- after
    the soak it runs itself on its own frames has passed: a form that diverges only after forty steps passes the few verified
    frames of a candidate run, fails the soak, and is never offered."""
    def cost(f):
        return 50.0 if "BTS_CVT_OVERLAP" in f else 40.0 if "BTS_FILTER_AHEAD" in f else 100.0
    monkeypatch.delenv("BTS_TUNE_ALLOW_CVT", raising=False)
    monkeypatch.setenv("BTS_TUNE_SOAK_PASSES", "0")              # no soak, no attestation: not offered
    r = _select(cost, monkeypatch)
    assert r["forms"] == "" and not any("CVT" in m.get("forms", "") or "FILTER_AHEAD" in m.get("forms", "") for m in r["table"]) and r["soak"] == []
    monkeypatch.setenv("BTS_TUNE_ALLOW_CVT", "1")                # attested by the caller
    r = _select(cost, monkeypatch)
    assert "BTS_FILTER_AHEAD=1" in r["forms"] and r["soak"] == []
    monkeypatch.delenv("BTS_TUNE_ALLOW_CVT")
    monkeypatch.setenv("BTS_TUNE_SOAK_PASSES", "8")              # the selection's own soak: 8 passes x 12 frames, both forms pass
    r = _select(cost, monkeypatch)
    assert [(x["name"], x["passed"], x["steps"]) for x in r["soak"]] == [("cvt_overlap", True, 96), ("filter_ahead", True, 96), ("smooth_deferred", True, 96)]
    assert "BTS_FILTER_AHEAD=1" in r["forms"]
    r = _select(cost, monkeypatch, late_wrong=("BTS_FILTER_AHEAD",))      # diverges at step 41: the soak sees it, twelve frames would not
    by = {x["name"]: x for x in r["soak"]}
    assert by["cvt_overlap"]["passed"] and not by["filter_ahead"]["passed"] and by["filter_ahead"]["mismatch_at"][0] == 51
    assert "BTS_FILTER_AHEAD" not in r["forms"] and "BTS_CVT_OVERLAP=1" in r["forms"]
    assert not any("FILTER_AHEAD" in m.get("forms", "") for m in r["table"])
    r = tune.select_forms(lambda forms: _Double(forms, cost), [(None, None, None)] * 12, warm=4, verify=3, repeats=1, clone=lambda fr: fr,
                          sync=lambda: None, skip=["soak:filter_ahead"])                # (blamed for a crash on an earlier attempt: skipped)
    assert {x["name"]: x.get("skipped", False) for x in r["soak"]} == {"cvt_overlap": False, "filter_ahead": True, "smooth_deferred": False}
    assert "FILTER_AHEAD" not in r["forms"]           # (the smooth variant uses BTS_FILTER_AHEAD too: every key it uses has to have passed)


def test_a_candidate_inside_the_noise_floor_is_no_difference(monkeypatch):
    """The default forms are measured twice (before and after the candidates); the spread of the two figures is the noise floor
    and a candidate has to beat max(margin, noise)."""
    calls = {"default": 0}

    def cost(f):
        if f == "":
            calls["default"] += 1
            return 100.0 if calls["default"] == 1 else 106.0        # the box drifted by 6 % between the two default runs
        return 96.0 if "BTS_LISTS_AHEAD" in f else 90.0 if "BTS_EDGES_SORTED" in f else 100.0
    r = _select(cost, monkeypatch)
    assert abs(r["noise_pct"] - 6.0) < 1e-6 and np.allclose(r["default_us_both"], [100.0, 106.0])
    by = {m["name"]: m for m in r["table"]}
    assert by["lists_ahead"]["verdict"] == "no difference"          # 4 % is more than the margin but inside the spread
    assert by["edges_sorted"]["verdict"] == "faster" and r["forms"] == "BTS_EDGES_SORTED=1"
    assert by["default(again)"]["identical"] is True


def test_nothing_faster_means_the_defaults(monkeypatch):
    r = _select(lambda f: 100.0, monkeypatch)
    assert r["forms"] == "" and r["chosen"] == "default"


needs_toolchain = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None, reason="needs ROCm's clang++ and make")


def _emu_env(tmp_path, **extra):
    subprocess.check_call(["make", "-C", EMU, "-s", "-j4"])
    site = tmp_path / "site"
    site.mkdir(exist_ok=True)
    (site / "sitecustomize.py").write_text("import tests.hip_emu.emu_prelude\n")
    env = dict(os.environ, BTS_LIB_PATH=os.path.join(EMU, "libbotsort_emu.so"),
               PYTHONPATH=os.pathsep.join([str(site), REPO] + ([os.environ["PYTHONPATH"]] if os.environ.get("PYTHONPATH") else [])))
    env.update(extra)
    return env


@needs_toolchain
def test_forms_string_wins_over_the_environment_on_the_product_abi(tmp_path):
    """bts_tracker_create_forms on the emulated build of the product sources: a key in the string beats the environment variable of
    the same name, "KEY=" switches a form off, an absent key falls back to the environment; bts_tracker_forms reports what is in
    effect (a form the configuration rules out -- sorted edges with the motion gate -- does not appear)."""
    code = r'''
import importlib, json, os
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
os.environ["BTS_LISTS_AHEAD"] = "1"
os.environ["BTS_LAP_FORM"] = "row"
def eff(forms=None, **kw):
    t = pkg.DeviceTracker(max_tracks=256, max_dets=64, feat_dim=128, forms=forms, **kw)
    f = t.forms(); t.close(); return f
out = {"env": eff(), "off": eff("BTS_LISTS_AHEAD=,BTS_LAP_FORM=col"), "mix": eff({"BTS_EDGES_SORTED": "1", "BTS_TILE_GOAL": "384", "BTS_LAP_HELPERS": "31"}),
       "gate": eff("BTS_EDGES_SORTED=1", motion_gate=1), "bare": eff("BTS_PREDICT_FIRST")}
print(json.dumps(out))
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=_emu_env(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["env"]["lists_ahead"] == "1" and o["env"]["lap_form"] == "row" and o["env"]["edges"] == "tiles"
    assert o["off"]["lists_ahead"] == "0" and o["off"]["lap_form"] == "col"
    assert o["mix"]["edges"] == "sorted" and o["mix"]["tile_goal"] == "384" and o["mix"]["lap_helpers"] == "31" and o["mix"]["lists_ahead"] == "1"
    assert o["gate"]["edges"] == "tiles"
    assert o["bare"]["predict_first"] == "1"


@needs_toolchain
def test_isolated_selection_survives_a_candidate_that_kills_the_child(tmp_path):
    """tune.run_isolated on the emulated library: three candidates, the second one takes the child process down (exit in the
    middle of the run).  The parent blames it from the progress lines, restarts the child without it, and the result carries
    the verdicts of the others: all bit-identical to the default forms on the verified frames."""
    env = _emu_env(tmp_path, BTS_TUNE_ONLY="lists_ahead,lap_row,predict_first", BTS_TUNE_CRASH_AT="lap_row")
    code = r'''
import importlib, json
tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
r = tune.run_isolated("c1", 1234, 0, n_frames=7, repeats=1, timeout_s=1200, tracker_args={"max_tracks": 128, "max_dets": 64, "feat_dim": 2048})
print(json.dumps(r))
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["crashed"] == ["lap_row"], o
    by = {m["name"]: m for m in o["table"]}
    assert by["lap_row"].get("skipped") is True and by["lap_row"]["eligible"] is False
    for name in ("default", "lists_ahead", "predict_first"):
        assert by[name]["identical"] is True and by[name]["errors"] == 0, by[name]
    assert "lap_row" not in o["forms"].lower() and "BTS_LAP_FORM" not in o["forms"]


def test_failing_default_forms_are_surfaced_not_replaced(monkeypatch):
    """The default forms raise a device error: that is a bug of the default path and the selection says so -- nothing is selected
    (a rescue form must not be adopted automatically); candidates that ran error-free are only listed."""
    monkeypatch.setattr(tune.time, "perf_counter", lambda: _Double.clock[0])
    frames = [(None, None, None)] * 12

    def make(forms):
        t = _Double(forms, lambda f: 100.0)
        t.err = "BTS_LAP_FORM=row" not in (forms or "")          # everything but the row-form solver fails
        return t
    r = tune.select_forms(make, frames, warm=4, verify=3, repeats=1, clone=lambda fr: fr, sync=lambda: None)
    assert r["forms"] == "" and r["chosen"] == "default" and r["default_errors"] == 1
    assert r["rescue_candidates"] == ["lap_row"] and "never adopted" in r["note"]
    assert {m["name"]: m for m in r["table"]}["default"]["errors"] == 1


@needs_toolchain
def test_isolated_selection_survives_a_candidate_that_hangs(tmp_path):
    """A candidate that never comes back (a kernel spinning on a flag nobody sets): the parent watches the child's progress
    lines, kills it after `candidate_timeout_s` of silence, blames the candidate that was running and restarts without it."""
    env = _emu_env(tmp_path, BTS_TUNE_ONLY="lists_ahead,lap_row", BTS_TUNE_HANG_AT="lists_ahead")
    code = r'''
import importlib, json
tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
r = tune.run_isolated("c1", 1234, 0, n_frames=7, repeats=1, timeout_s=1200, candidate_timeout_s=30,
                      tracker_args={"max_tracks": 128, "max_dets": 64, "feat_dim": 2048})
print(json.dumps(r))
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["crashed"] == ["lists_ahead"] and not o.get("partial"), o
    by = {m["name"]: m for m in o["table"]}
    assert by["lists_ahead"].get("skipped") is True
    assert by["default"]["identical"] is True and by["lap_row"]["identical"] is True and by["lap_row"]["errors"] == 0


@needs_toolchain
def test_the_selections_own_soak_runs_the_side_stream_forms_on_the_emulated_library(tmp_path):
    """Round 6: tune.soak -- what select_forms runs before it offers BTS_CVT_OVERLAP / BTS_FILTER_AHEAD -- on the emulated product
    library: 2 passes over 8 frames of a 48 x 24 x 128 stream with churn, digests every 5th step, the last pass without the
    residency promise.  The form set returns the default forms' digests and final state; the record says how many steps and digests
    that took."""
    env = _emu_env(tmp_path)
    code = r'''
import importlib, json, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
st = pkg.synth.SyntheticStream(48, 24, 128, seed=11, churn=0.1)
frames = []
for _ in range(8):
    fr = st.next()
    frames.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()))
make = lambda forms: pkg.DeviceTracker(max_tracks=128, max_dets=64, feat_dim=128, forms=forms or None)
out = [tune.soak(make, frames, forms, passes=2, stride=5) for _, _, forms in tune.SOAK_FORMS[-1:]]      # (the most complete set: ahead + deferred smooth)
print(json.dumps(out))
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert [o["forms"] for o in out] == ["BTS_FILTER_AHEAD=1,BTS_SMOOTH_DEFERRED=1"]
    for o in out:
        assert o["passed"] and o["steps"] == 16 and o["digests"] == 4 and o["mismatch_at"] == [] and o["errors"] == [0, 0] and o["state_equal"], o
