"""Selection of a tracker's EXECUTION FORMS on the device it runs on (include/botsort_hip.h, bts_tracker_create_forms).

Several kernels / launch structures of a step exist in more than one form with bit-identical results; which is faster
depends on the table size, the scene and the part.  `select_forms` does what a find-mode does for a convolution: it steps
the caller's own frames through one tracker per candidate and keeps a candidate only if

  * every checked frame's RESULT is identical to the default forms' -- assignment vectors of the three stages, both track
    lists in order (ids, states, frame counters, detection indices), Kalman means, boxes and scores, bit for bit -- and the
    device raised no error, and
  * it is faster than the default forms by more than `margin`.

The winners of the groups are combined; the combination is measured and verified like any candidate and kept only if it
beats the best single change.  Nothing here is a fallback: every candidate is the HIP library (no CPU path), and the
default forms are what `DeviceTracker()` uses when no selection was made.

`run_isolated` does the same in a child process (`python tune.py ...`), so that a candidate that takes the process down
-- the forms built while no GPU was available have never met one -- costs the candidate, not the caller: the parent reads
the child's progress lines, blames the candidate that was running, and restarts the child without it.  bench.py uses that
(`--forms auto`) in its untimed set-up, next to the generation of its frames.

Reference seam: the forms live entirely below `BoTSORT.update` (D:1414-1639); the reference has no counterpart.
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

# (group, name, forms): groups are alternatives of ONE switch -- at most one member of a group is taken
CANDIDATES: Tuple[Tuple[str, str, Dict[str, str]], ...] = (
    ("lists_ahead", "lists_ahead", {"BTS_LISTS_AHEAD": "1"}),
    ("lists_ahead", "lists_ahead_predict_front", {"BTS_LISTS_AHEAD": "1", "BTS_PREDICT_FRONT": "1"}),
    ("merge_early", "merge_early", {"BTS_MERGE_EARLY": "1"}),
    ("renorm_fused", "renorm_fused", {"BTS_RENORM_FUSED": "1"}),
    ("edges", "edges_sorted", {"BTS_EDGES_SORTED": "1"}),
    ("edges", "edges_grid", {"BTS_EDGES_GRID": "1"}),
    ("predict_first", "predict_first", {"BTS_PREDICT_FIRST": "1"}),
    ("shadow_planes", "shadow_planes", {"BTS_SHADOW_PLANES": "1"}),
    ("filter_guess", "filter_guess", {"BTS_FILTER_GUESS": "1"}),
    ("filter_pw", "filter_pw_768", {"BTS_FILTER_PW": "768"}),       # more workgroups than resident slots: finer tail, colder tile openings
    ("filter_pw", "filter_pw_2048", {"BTS_FILTER_PW": "2048"}),     # (one tile per workgroup at the headline size: the hardware's own backfill)
    # tile shape of the one-pass pre-filter (k_filter_t, round 5): 160 x 128 = two tile rounds instead of three at the headline size;
    # 128 x 64 = three resident workgroups per CU
    ("filter_tile", "filter_tile_160x128", {"BTS_FILTER_TILE": "160x128"}),
    ("filter_tile", "filter_tile_128x64", {"BTS_FILTER_TILE": "128x64"}),
    ("filter_first16", "filter_first16_5", {"BTS_I8_FIRST16": "5"}),
    ("filter_first16", "filter_first16_7", {"BTS_I8_FIRST16": "7"}),
    ("lap_self", "lap_self", {"BTS_LAP_SELF": "1"}),
    ("lap_form", "lap_row", {"BTS_LAP_FORM": "row"}),
    ("lap_helpers", "lap_helpers_31", {"BTS_LAP_HELPERS": "31"}),
    ("lap_helpers", "lap_helpers_127", {"BTS_LAP_HELPERS": "127"}),
    ("chunks", "lists_unsplit", {"BTS_CHUNK_ROWS": "1000000"}),
    ("chunks", "lists_chunks_4096", {"BTS_CHUNK_ROWS": "4096"}),
    ("chunks", "lists_chunks_1024", {"BTS_CHUNK_ROWS": "1024"}),
    ("tile_goal", "tile_goal_384", {"BTS_TILE_GOAL": "384"}),
    ("tile_goal", "tile_goal_1536", {"BTS_TILE_GOAL": "1536"}),
)

# BTS_CVT_OVERLAP is NOT offered unconditionally: its side-stream ordering (two events per frame) is the one thing neither the CPU
# emulator (events are no-ops there) nor a digest of five frames can vouch for -- a race detector with five samples (VERDICT r4,
# weak 2).  It becomes a candidate only once a digest soak has passed on the device: either the 2 000-frame one of
# scripts/soak_cvt_overlap.py, attested by the caller (BTS_TUNE_ALLOW_CVT=1), or -- round 6 -- the one select_forms runs itself on
# its own frames before it offers these forms (soak() below; BTS_TUNE_SOAK_PASSES, default 8 passes over the frames, digests every
# 17th step and no other host synchronisation).  (Forms that only pay TOGETHER are offered together: the conversion leaves the
# first launch for nothing while the list jobs still sit in it.)
CVT_CANDIDATES: Tuple[Tuple[str, str, Dict[str, str]], ...] = (
    ("lists_ahead", "lists_ahead_cvt", {"BTS_LISTS_AHEAD": "1", "BTS_CVT_OVERLAP": "1"}),
    ("lists_ahead", "lists_ahead_cvt_predict_front", {"BTS_LISTS_AHEAD": "1", "BTS_CVT_OVERLAP": "1", "BTS_PREDICT_FRONT": "1"}),
    ("cvt_overlap", "cvt_overlap", {"BTS_CVT_OVERLAP": "1"}),
    # round 6, opt-in THROUGHPUT mode: the pre-filter of frame f on the side stream ahead of step f (main pass over the table's slots
    # + det x det pass; implies lists-ahead, predict-front and the side-stream conversion).  Same kind of cross-stream ordering,
    # same gate: offered after `scripts/soak_cvt_overlap.py 2000 BTS_FILTER_AHEAD=1` has passed on the device.  It pays only
    # when the caller steps frames back to back (the host ahead of the device), which is how the selection times its candidates.
    ("lists_ahead", "filter_ahead", {"BTS_FILTER_AHEAD": "1"}),
    ("lists_ahead", "filter_ahead_merge_early", {"BTS_FILTER_AHEAD": "1", "BTS_MERGE_EARLY": "1"}),
    ("lists_ahead", "filter_ahead_pw_384", {"BTS_FILTER_AHEAD": "1", "BTS_FILTER_AHEAD_PW": "384"}),      # (default: 256 side workgroups, one per CU)
    ("lists_ahead", "filter_ahead_pw_512", {"BTS_FILTER_AHEAD": "1", "BTS_FILTER_AHEAD_PW": "512"}),
    # + the `smooth` EMA of the rewritten tracks on a third stream behind the step (half of tk_mid2's bytes leave the step's stream)
    ("lists_ahead", "filter_ahead_smooth", {"BTS_FILTER_AHEAD": "1", "BTS_SMOOTH_DEFERRED": "1"}),
    ("lists_ahead", "filter_ahead_smooth_merge_early", {"BTS_FILTER_AHEAD": "1", "BTS_SMOOTH_DEFERRED": "1", "BTS_MERGE_EARLY": "1"}),
    # + the sorted stage-1 edges reading the side passes' candidate lists (rows translated from slot / previous-detection space)
    ("lists_ahead", "filter_ahead_sorted_smooth", {"BTS_FILTER_AHEAD": "1", "BTS_SMOOTH_DEFERRED": "1", "BTS_EDGES_SORTED": "1"}),
    # everything DESIGN.md 3.3's budget counts on, in one candidate (the greedy combination of single winners would also get here)
    ("lists_ahead", "throughput_all", {"BTS_FILTER_AHEAD": "1", "BTS_SMOOTH_DEFERRED": "1", "BTS_EDGES_SORTED": "1", "BTS_MERGE_EARLY": "1",
                                       "BTS_RENORM_FUSED": "1", "BTS_LAP_SELF": "1"}),
)


# The side-stream forms and the key that marks their candidates.  Their evidence is a SOAK (below): since round 6 the selection runs
# it itself, on the device, before it offers them -- no human sets BTS_TUNE_ALLOW_CVT=1 after reading a log.
SOAK_FORMS: Tuple[Tuple[str, str, Dict[str, str]], ...] = (
    ("cvt_overlap", "BTS_CVT_OVERLAP", {"BTS_CVT_OVERLAP": "1", "BTS_LISTS_AHEAD": "1"}),
    ("filter_ahead", "BTS_FILTER_AHEAD", {"BTS_FILTER_AHEAD": "1"}),
    ("smooth_deferred", "BTS_SMOOTH_DEFERRED", {"BTS_FILTER_AHEAD": "1", "BTS_SMOOTH_DEFERRED": "1"}),
)


def default_candidates():
    return CANDIDATES + (CVT_CANDIDATES if os.environ.get("BTS_TUNE_ALLOW_CVT") == "1" else ())


def soak(make_tracker: Callable[[str], object], frames: Sequence[tuple], forms: Dict[str, str], passes: int = 8, stride: int = 17,
         clone: Callable = None, sync: Callable = None, progress: Callable[[], None] = None) -> dict:
    """Digest equality of `forms` against the default forms over passes x len(frames) steps: both trackers step the same frames
    back to back, the host synchronises with the device only where a digest is taken -- every `stride`-th step and the last --, so
    the form's side stream really runs under the previous steps' kernels; the last pass runs WITHOUT the residency promise (the
    side stream is then ordered behind the caller's stream at every step).  The frames are re-cloned per pass (the step
    renormalises used rows in place, Q8) and stepped again: not a video, but identical inputs for both trackers, which is all an
    equality test needs.  -> {"steps", "digests", "mismatch_at", "errors", "state_equal", "passed"}.  What
    scripts/soak_cvt_overlap.py does by hand, sized to run inside a bench."""
    import torch
    clone = clone or (lambda fr: tuple(x.clone() if i >= 2 and x is not None else x for i, x in enumerate(fr)))
    sync = sync or torch.cuda.synchronize
    ref, exp = make_tracker(""), make_tracker(forms_text(forms))
    bad: List[int] = []
    checked = step = 0
    try:
        for p in range(max(1, passes)):
            resident = p + 1 < max(1, passes) or passes == 1
            if hasattr(exp, "inputs_resident"):
                exp.inputs_resident(resident)
            a, b = [clone(fr) for fr in frames], [clone(fr) for fr in frames]
            sync()
            if progress:
                progress()               # (one sign of life per pass: run_isolated's parent takes silence for a hang)
            for fa, fb in zip(a, b):
                ref.step(*fa)
                exp.step(*fb)
                step += 1
                if step % stride == 0:
                    checked += 1
                    if _digest(ref) != _digest(exp):
                        bad.append(step)
            sync()
        checked += 1
        if _digest(ref) != _digest(exp):
            bad.append(step)
        state_equal = _state_digest(ref) == _state_digest(exp)
        errors = (int(ref.counts()["errors"]), int(exp.counts()["errors"]))
    finally:
        ref.close()
        exp.close()
    return {"forms": forms_text(forms), "steps": step, "digests": checked, "mismatch_at": bad[:8], "errors": list(errors), "state_equal": state_equal,
            "passed": bool(not bad and errors == (0, 0) and state_equal)}


def forms_text(forms: Dict[str, str]) -> str:
    return ",".join("%s=%s" % kv for kv in sorted(forms.items()))


def _digest(trk) -> str:
    """Everything a caller can observe of the frame just stepped, hashed (host synchronisation: verification frames only)."""
    h = hashlib.blake2b(digest_size=16)
    for k in (1, 2, 3):
        r = trk.stage_result(k)
        h.update(r["x"].tobytes())
        h.update(r["y"].tobytes())
    for which in (0, 1):
        t = trk.export(which)
        for key in ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len", "det_index", "mean", "tlbr", "score"):
            h.update(t[key].tobytes())
    return h.hexdigest()


def _state_digest(trk) -> str:
    """What the frame digest does not see and several forms rewrite: covariances and body_curr / body_smooth rows of
    both lists, in list order.  Large (tens of MB at the headline size): taken once per run, on the final frame."""
    h = hashlib.blake2b(digest_size=16)
    if not hasattr(trk, "export_state"):
        return ""
    for which in (0, 1):
        st = trk.export_state(which, cov=True, body=True, face=False)
        for key in ("covariance", "body_curr_feature", "body_smooth_feature"):
            h.update(st[key].tobytes())
    return h.hexdigest()


def measure(make_tracker: Callable[[str], object], frames: Sequence[tuple], forms: str, warm: int, verify: int,
            clone: Callable = None, sync: Callable = None) -> dict:
    """One tracker with `forms` through `frames`: frames [0, warm) untimed (the first `verify` of them digested), the rest
    timed back to back without any host synchronisation in between, the last frame digested again.  Frames are (boxes, scores,
    feats[, face_feats]) device tensors; the feature rows are CLONED per run (the step renormalises used rows in place, Q8)."""
    import torch
    clone = clone or (lambda fr: tuple(x.clone() if i >= 2 and x is not None else x for i, x in enumerate(fr)))
    sync = sync or torch.cuda.synchronize
    mine = [clone(fr) for fr in frames]
    sync()                                   # (the clones above are asynchronous copies: the residency promise below is only true
                                             #  once they have completed -- ADVICE r4)
    trk = make_tracker(forms)
    try:
        if hasattr(trk, "inputs_resident"):
            trk.inputs_resident(True)        # (the frames are complete in device memory before the loop that steps on them)
        digests = []
        for k in range(warm):
            trk.step(*mine[k])
            if k < verify:
                digests.append(_digest(trk))
        sync()
        t0 = time.perf_counter()
        for fr in mine[warm:]:
            trk.step(*fr)
        sync()
        t1 = time.perf_counter()
        digests.append(_digest(trk))
        digests.append(_state_digest(trk))
        errors = int(trk.counts()["errors"])
        effective = trk.forms() if hasattr(trk, "forms") else {}
    finally:
        trk.close()
    n = max(1, len(mine) - warm)
    return {"us_per_step": 1e6 * (t1 - t0) / n, "digests": digests, "errors": errors, "effective": effective, "timed_steps": n}


def select_forms(make_tracker: Callable[[str], object], frames: Sequence[tuple], candidates=None, warm: int = 6, verify: int = 4,
                 margin: float = 0.02, skip: Sequence[str] = (), report: Callable[[dict], None] = None, repeats: int = 3,
                 clone: Callable = None, sync: Callable = None) -> dict:
    """-> {"forms": "KEY=VALUE,..." ("" = the defaults), "default_us", "chosen_us", "noise_pct", "table": [...]}.  `report` receives
    a progress record before and after every candidate (run_isolated's crash attribution).

    Statistics: every figure is the MINIMUM of `repeats` (default 3) runs of all timed frames (callers pass
    >= 100 timed frames, or >= 25 ms of them); the default forms are measured twice -- before the first and after the last
    candidate -- and the spread of those two figures is the noise floor (`noise_pct`): a candidate counts as faster only beyond
    max(margin, noise), anything inside is "no difference" and the default stays."""
    report = report or (lambda rec: None)
    soaks: List[dict] = []
    if candidates is None:
        candidates = default_candidates()
        # the side-stream forms (BTS_CVT_OVERLAP, BTS_FILTER_AHEAD): offered once their soak has passed -- attested by the caller
        # (BTS_TUNE_ALLOW_CVT=1, after scripts/soak_cvt_overlap.py) or run HERE, on these frames (BTS_TUNE_SOAK_PASSES, default 8
        # passes; 0 = never).  A soak that takes the child down is blamed like a candidate ("soak:<name>") and skipped on the restart.
        passes = int(os.environ.get("BTS_TUNE_SOAK_PASSES", "8") or 0)
        only0 = [n for n in os.environ.get("BTS_TUNE_ONLY", "").split(",") if n]
        if os.environ.get("BTS_TUNE_ALLOW_CVT") != "1" and passes > 0 and len(frames) >= 8:
            for sname, key, forms in SOAK_FORMS:
                if only0 and not any(c[1] in only0 for c in CVT_CANDIDATES if key in c[2]):
                    continue                 # (a subset of the candidates was asked for and none of this form's is in it)
                if "soak:" + sname in skip:
                    soaks.append({"name": sname, "skipped": True, "passed": False})
                    continue
                report({"event": "begin", "name": "soak:" + sname})
                try:
                    r = soak(make_tracker, frames, forms, passes=passes, clone=clone, sync=sync,
                             progress=lambda n=sname: report({"event": "tick", "name": "soak:" + n}))
                except Exception as e:           # (a library without the form)
                    r = {"passed": False, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
                r["name"] = sname
                soaks.append(r)
                report(dict(event="end", name="soak:" + sname, **{k: v for k, v in r.items() if k != "name"}))
            # a gated candidate is offered iff EVERY side-stream key it uses has passed its soak
            key_of = {sname: key for sname, key, _ in SOAK_FORMS}
            passed = {key_of[r["name"]] for r in soaks if r.get("passed")}
            keys = {key for _, key, _ in SOAK_FORMS}
            candidates = tuple(candidates) + tuple(c for c in CVT_CANDIDATES if (keys & set(c[2])) and (keys & set(c[2])) <= passed and c not in candidates)
    only = [n for n in os.environ.get("BTS_TUNE_ONLY", "").split(",") if n]      # (tests: a subset of the candidates by name)
    if only:
        candidates = [c for c in candidates if c[1] in only]
    assert len(frames) > warm >= verify >= 1, "need more frames than warm-up frames"

    def run(name, forms):
        report({"event": "begin", "name": name})
        best = None
        for _ in range(max(1, repeats)):
            m = measure(make_tracker, frames, forms_text(forms), warm, verify, clone, sync)
            if best is None or m["us_per_step"] < best["us_per_step"]:
                best = m
        best.update(name=name, forms=forms_text(forms))
        return best

    ref = run("default", {})
    ref["identical"], ref["eligible"] = True, True
    report(dict(event="end", **{k: v for k, v in ref.items() if k != "digests"}))
    table = [ref]
    if ref["errors"]:
        # The default forms themselves raised a device error (a solver hand-over that timed out, a list job's look-back ...): that
        # is a correctness bug of the default path and it is SURFACED, not papered over by switching kernels: nothing is
        # selected.  The candidates are still run so that the report says which forms complete error-free -- a debugging aid
        # ("rescue_candidates"), never adopted automatically.
        rescue = []
        for group, name, forms in candidates:
            if name in skip:
                continue
            try:
                m = run(name, forms)
            except Exception as e:
                table.append({"name": name, "forms": forms_text(forms), "error": "%s: %s" % (type(e).__name__, str(e)[:200]), "eligible": False})
                continue
            m["identical"], m["eligible"] = None, False
            report(dict(event="end", **{k: v for k, v in m.items() if k != "digests"}))
            table.append(m)
            if m["errors"] == 0:
                rescue.append(name)
                if len(rescue) >= 3:
                    break
        return {"forms": "", "chosen": "default", "default_us": ref["us_per_step"], "chosen_us": ref["us_per_step"], "table": _strip(table),
                "default_errors": ref["errors"], "rescue_candidates": rescue, "soak": soaks,
                "note": "the DEFAULT forms raised a device error on these frames: nothing selected (candidates that ran error-free are "
                        "listed in rescue_candidates; they are unverified and never adopted automatically)"}

    def judge(m):
        m["identical"] = m["digests"] == ref["digests"]
        m["eligible"] = bool(m["identical"] and m["errors"] == 0)
        m["speedup"] = ref["us_per_step"] / m["us_per_step"] if m["us_per_step"] > 0 else 0.0
        report(dict(event="end", **{k: v for k, v in m.items() if k != "digests"}))
        table.append(m)
        return m

    measured: List[dict] = []
    for group, name, forms in candidates:
        if name in skip:
            table.append({"name": name, "forms": forms_text(forms), "skipped": True, "eligible": False})
            continue
        try:
            m = judge(run(name, forms))
        except Exception as e:           # (a form this library / configuration does not have: reported, never selected)
            table.append({"name": name, "forms": forms_text(forms), "error": "%s: %s" % (type(e).__name__, str(e)[:200]), "eligible": False})
            report({"event": "end", "name": name, "error": str(e)[:200]})
            continue
        m["group"] = group
        measured.append(m)
    # the default forms once more, behind the candidates: the two figures' spread is what this box's timing is worth today
    ref_b = run("default", {})
    ref_b["name"] = "default(again)"
    ref_b["identical"], ref_b["eligible"] = ref_b["digests"] == ref["digests"], False
    report(dict(event="end", **{k: v for k, v in ref_b.items() if k != "digests"}))
    table.append(ref_b)
    lo, hi = sorted((ref["us_per_step"], ref_b["us_per_step"]))
    noise = (hi - lo) / lo if lo > 0 else 0.0
    ref_us = lo
    need = 1.0 + max(margin, noise)
    winners: Dict[str, dict] = {}
    for m in measured:
        m["speedup"] = ref_us / m["us_per_step"] if m["us_per_step"] > 0 else 0.0
        m["verdict"] = ("not identical" if not m["identical"] else "device error" if m["errors"] else
                        "faster" if m["speedup"] > need else "slower" if m["speedup"] < 2.0 - need else "no difference")
        if m["verdict"] == "faster" and (m["group"] not in winners or m["us_per_step"] < winners[m["group"]]["us_per_step"]):
            winners[m["group"]] = m
    # combine greedily: the best single change first, then the other winners by gain -- each kept only if the combination is
    # again identical, error-free and faster than what is held so far by more than half the threshold
    chosen = dict(ref, us_per_step=ref_us)
    order = sorted(winners.values(), key=lambda m: m["us_per_step"])
    held: Dict[str, str] = {}
    names: List[str] = []
    for m in order:
        forms = dict(held)
        forms.update(dict(kv.split("=", 1) for kv in m["forms"].split(",")))
        if not names:
            held, names, chosen = forms, [m["name"]], m
            continue
        name = "+".join(names + [m["name"]])
        if name in skip:
            continue
        try:
            combo = judge(run(name, forms))
        except Exception as e:
            table.append({"name": name, "forms": forms_text(forms), "error": "%s: %s" % (type(e).__name__, str(e)[:200]), "eligible": False})
            continue
        combo["speedup"] = ref_us / combo["us_per_step"] if combo["us_per_step"] > 0 else 0.0
        if combo["eligible"] and combo["us_per_step"] < chosen["us_per_step"] / (1.0 + (need - 1.0) / 2):
            held, chosen = forms, combo
            names.append(m["name"])
    timed = ref.get("timed_steps", 0)
    return {"forms": chosen["forms"], "chosen": chosen["name"], "default_us": ref_us, "chosen_us": chosen["us_per_step"],
            "default_us_both": [ref["us_per_step"], ref_b["us_per_step"]], "noise_pct": 100.0 * noise, "needed_speedup": need,
            "table": _strip(table), "soak": soaks, "frames": len(frames), "warm": warm, "timed_steps": timed, "repeats": repeats,
            "thin": bool(timed < 100 and timed * ref_us < 25e3),
            "verified_frames": verify + 1, "margin": margin,
            "rule": "a candidate is eligible only if the digests of its verified frames (assignment vectors of the three stages, both "
                    "lists in order, Kalman means, boxes, scores; on the final frame also covariances and body feature rows) equal the "
                    "default forms' and the device raised no error; it is taken only if it is faster than the default forms (min of "
                    "%d runs each; the default measured before and after) by more than max(margin, noise)" % max(1, repeats)}


def _strip(table):
    return [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in m.items() if k != "digests"} for m in table]


# ------------------------------------------------------------------------------------------- isolated (child process)
def run_isolated(workload: str, seed: int, device: int = 0, n_frames: int = 116, tracker_args: Optional[dict] = None,
                 timeout_s: float = 600.0, max_restarts: int = 3, python: str = None, extra_env: Optional[dict] = None, repeats: int = 3,
                 candidate_timeout_s: float = 90.0, startup_timeout_s: float = 300.0, margin: float = 0.02) -> dict:
    """select_forms on `n_frames` of the seeded synthetic workload in a child process.  The child's progress lines are read as
    they come: a candidate that kills the child, or keeps it silent for more than `candidate_timeout_s` (a hang), is blamed and
    skipped on the restart (the child's start -- interpreter, frames -- gets `startup_timeout_s`).  Returns select_forms'
    dictionary plus "crashed": [names]; if the selection cannot be completed within `timeout_s`, the best single candidate that
    WAS measured identical, error-free and faster is returned (marked "partial"), else the defaults."""
    import queue
    import threading
    crashed: List[str] = []
    last = None
    measured: Dict[str, dict] = {}          # name -> its "end" record, over all restarts
    deadline = time.time() + timeout_s
    for _ in range(max_restarts + 1):
        cmd = [python or sys.executable, os.path.abspath(__file__), "--workload", workload, "--seed", str(seed), "--device", str(device),
               "--frames", str(n_frames), "--repeats", str(repeats), "--margin", str(margin), "--tracker-args", json.dumps(tracker_args or {})] + (["--skip", ",".join(crashed)] if crashed else [])
        env = dict(os.environ, **(extra_env or {}))
        running, result, hung = None, None, False
        try:
            p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, bufsize=1)
            q: "queue.Queue" = queue.Queue()
            err_tail: List[str] = []

            def pump(stream, sink):
                for ln in stream:
                    sink(ln)
                sink(None)
            threading.Thread(target=pump, args=(p.stdout, q.put), daemon=True).start()
            threading.Thread(target=pump, args=(p.stderr, lambda ln: err_tail.append(ln) if ln else None), daemon=True).start()
            quiet_since, limit = time.time(), startup_timeout_s
            while True:
                try:
                    ln = q.get(timeout=1.0)
                except queue.Empty:
                    now = time.time()
                    if now - quiet_since > limit or now >= deadline:
                        hung = True
                        p.kill()
                        break
                    continue
                if ln is None:
                    break
                if not ln.startswith("{"):
                    continue
                try:
                    rec = json.loads(ln)
                except ValueError:
                    continue
                quiet_since, limit = time.time(), candidate_timeout_s * max(1, repeats)
                if rec.get("event") == "begin":
                    running = rec["name"]
                elif rec.get("event") == "end":
                    running = None
                    if "name" in rec:
                        measured[rec["name"]] = rec
                elif rec.get("event") == "result":
                    result = rec["result"]
            p.wait(timeout=30)
            last = {"rc": p.returncode, "hung": hung, "stderr_tail": "".join(err_tail)[-400:]}
        except Exception as e:
            last = {"rc": None, "stderr_tail": "%s: %s" % (type(e).__name__, e)}
        if result is not None:
            result["crashed"] = crashed
            return result
        if running is None or running == "default" or running in crashed or time.time() >= deadline:
            break                          # (nothing to blame, or the default forms themselves failed, or out of time)
        crashed.append(running)
    # not completed: whatever was measured still counts
    ref = measured.get("default")
    best = None
    if ref and not ref.get("errors"):
        for name, m in measured.items():
            if name != "default" and m.get("eligible") and m.get("us_per_step", 0) > 0 and ref["us_per_step"] / m["us_per_step"] > 1.0 + margin:
                if best is None or m["us_per_step"] < best["us_per_step"]:
                    best = m
    out = {"forms": best["forms"] if best else "", "chosen": best["name"] if best else "default", "crashed": crashed, "partial": True,
           "table": list(measured.values()), "child": last,
           "note": "form selection did not complete" + (": the best single candidate measured so far is used" if best else ": the defaults are used")}
    if ref:
        out["default_us"] = ref.get("us_per_step")
        out["chosen_us"] = (best or ref).get("us_per_step")
    return out


def _main():
    import argparse
    import importlib
    ap = argparse.ArgumentParser(description="execution-form selection on a synthetic workload (child of tune.run_isolated / bench.py)")
    ap.add_argument("--workload", default="target")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--frames", type=int, default=116)
    ap.add_argument("--warm", type=int, default=6)
    ap.add_argument("--verify", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.02)
    ap.add_argument("--skip", default="")
    ap.add_argument("--tracker-args", default="{}")
    a = ap.parse_args()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import torch
    assert torch.cuda.is_available(), "form selection runs on the device it selects for"
    torch.cuda.set_device(a.device % max(1, torch.cuda.device_count()))
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    K, M, D = pkg.synth.workload_dims(a.workload)
    st = pkg.synth.make_workload(a.workload, seed=a.seed)
    frames = []
    for _ in range(a.frames):
        fr = st.next()
        frames.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()))
    targs = json.loads(a.tracker_args)
    targs.setdefault("max_tracks", K + max(64, K // 8))
    targs.setdefault("max_dets", max(K, M))
    targs.setdefault("feat_dim", D)

    def make(forms):
        return pkg.DeviceTracker(device=torch.cuda.current_device(), forms=forms or None, **targs)

    def report(rec):
        print(json.dumps(rec), flush=True)
        if rec.get("event") == "begin" and rec.get("name") == os.environ.get("BTS_TUNE_CRASH_AT"):
            os._exit(9)                  # (tests/test_tune_host.py: a candidate that takes the process down)
        if rec.get("event") == "begin" and rec.get("name") == os.environ.get("BTS_TUNE_HANG_AT"):
            time.sleep(1e6)              # (... and one that hangs)

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        res = select_forms(make, frames, warm=a.warm, verify=a.verify, skip=[s for s in a.skip.split(",") if s], report=report, repeats=a.repeats, margin=a.margin)
    print(json.dumps({"event": "result", "result": res}), flush=True)


if __name__ == "__main__":
    _main()
