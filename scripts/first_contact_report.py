#!/usr/bin/env python
"""Turn the outputs of the first GPU batches (scripts/gpu_r5_a.sh, gpu_r5_b.sh -> gpurun_out/r5a, r5b) or of ANY bench.py line
(BENCH_rNN.json of the driver included) into the tables the next decisions need: per-kernel time against DESIGN.md section 3.3's
budget, the form selection's verdicts, the form matrix (mean of the repetitions, delta to the default forms), the trace variants'
phase tables.  CPU only.

    python scripts/first_contact_report.py [gpurun_out/r5a/bench_driver.json | BENCH_r05.json ...] [--matrix gpurun_out/r5b]"""
import glob
import json
import os
import re
import sys

# (round 2's measured figure, the budget of DESIGN.md 3.3 -- with BTS_FILTER_AHEAD k_filter is not on the step's stream at all)
BUDGET = {"k_filter": (59.7, 0), "k_edges1": (27.4, 28), "k_lap": (22.4, 17.4), "prep": (17.8, 7), "k_mid2": (17.6, 17.6),
          "k_apply23": (6.1, 11.6), "merge": (10.5, 0), "dedupe": (11.3 + 5.4, 11.3 + 10), "step": (193.7, 117)}


def load_line(path):
    txt = open(path).read()
    try:
        d = json.loads(txt)
    except ValueError:
        d = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
    return d.get("parsed") or d          # (the driver's BENCH_rNN.json wraps the line)


def line_report(path):
    d = load_line(path)
    if not d:
        print("## %s: no parsed line" % path)
        return
    print("## %s" % path)
    print("value %.0f steps/s (%s) | default forms %s | selected forms %s | %.1f us/step | workload %s"
          % (d["value"], d.get("value_is"), d.get("value_default_forms"), d.get("value_selected_forms"), 1e3 * d["ms_per_step"], d["config"]["workload"][:60]))
    for k in ("invalid", "truncated", "device_errors_after_timed_region"):
        if d.get(k):
            print("!! %s: %s" % (k, d[k]))
    im = d.get("id_match") or {}
    print("id_match: %s / %s frames identical, pairs %s / %s, kalman err %s" % (im.get("frames_identical"), im.get("frames"), im.get("assignment_pairs_equal"), im.get("assignment_pairs"), im.get("kalman_mean_max_abs_err")))
    r = d.get("roofline") or {}
    print("roofline: %s avg %s us, frac %s of %s; step frac_survey %s" % (r.get("kernel"), r.get("avg_us"), r.get("frac"), r.get("bound_peak"), (r.get("step") or {}).get("frac_survey")))
    seg = d.get("segments_us") or {}
    if seg:
        print("\n| segment | round 2 (us) | now (us) | budget |\n|---|---|---|---|")
        for k, (old, bud) in BUDGET.items():
            if k in seg:
                print("| %s | %.1f | %.1f | %s |" % (k, old, seg[k], bud))
        print("(event pairs add ~4 us per segment; `step` is the whole step with one pair)")
    oth = d.get("segments_us_other_forms")
    if oth and seg:
        print("\nper kernel, forms of `value` vs the other forms measured (%r):" % oth.get("forms"))
        print("| segment | value's forms | other forms | delta |\n|---|---|---|---|")
        for k, v in oth["segments_us"].items():
            if k in seg:
                print("| %s | %.1f | %.1f | %+.1f |" % (k, seg[k], v, seg[k] - v))
    fs = d.get("form_selection") or {}
    if fs.get("table"):
        print("\nform selection: chosen %r, default %s us (both %s, noise %.2f %%), needed speed-up %.3f" % (fs.get("forms"), fs.get("default_us"), fs.get("default_us_both"), fs.get("noise_pct") or 0.0, fs.get("needed_speedup") or 0.0))
        print("| candidate | us/step | vs default | identical | errors | verdict |\n|---|---|---|---|---|---|")
        for m in fs["table"]:
            print("| %s | %s | %s | %s | %s | %s |" % (m.get("name"), m.get("us_per_step"), ("%.3f" % m["speedup"]) if m.get("speedup") else "", m.get("identical"), m.get("errors", m.get("error", "")), m.get("verdict", "skipped" if m.get("skipped") else "")))
        for sk in fs.get("soak") or []:          # (round 6: the side-stream forms' soak, run by the selection itself)
            print("  soak %-14s %s  (%s steps, %s digests, mismatches at %s, device errors %s, final state %s)"
                  % (sk.get("name"), "PASS" if sk.get("passed") else "FAIL" if not sk.get("skipped") else "skipped (blamed for a crash)", sk.get("steps"), sk.get("digests"),
                     sk.get("mismatch_at") or "none", sk.get("errors"), "equal" if sk.get("state_equal") else sk.get("error") or "DIFFERENT"))
        if fs.get("crashed"):
            print("crashed / hung candidates:", fs["crashed"])
    pf = d.get("prefilter") or {}
    if pf:
        print("\nprefilter: tile %s, tiles %s, K stages %s (full: %s), prefix exits %s, candidates %s" % (pf.get("tile"), pf.get("tiles"), pf.get("k_stages"), pf.get("full_k_stages_equivalent"), pf.get("prefix_exits"), pf.get("candidates")))
    for key, t in (d.get("trace_variants") or {}).items():
        print("\n### trace variant: %s" % key)
        for ln in (t.get("table") or [t.get("error") or t.get("stderr_tail") or str(t)]):
            print("    " + str(ln))
    cb = d.get("cpu_baseline") or {}
    if cb:
        print("\ncpu_baseline: %s steps/s faithful, vectorised %s, %s cores used of %s, %s" % (cb.get("value"), (cb.get("vectorised") or {}).get("value"), cb.get("cores"), cb.get("host_cores"), cb.get("cpu_model")))
    print()


def matrix_report(folder):
    rows = {}
    for f in sorted(glob.glob(os.path.join(folder, "opt_*_*.json"))):
        m = re.match(r"opt_(.*)_(\d+)\.json", os.path.basename(f))
        try:
            d = load_line(f)
            rows.setdefault(m.group(1), []).append(d)
        except Exception:
            rows.setdefault(m.group(1), [])
    if not rows:
        return
    base = rows.get("default") or []
    b_us = min((1e3 * d["ms_per_step"] for d in base), default=None)
    print("## form matrix (%s): us/step = min over repetitions; segments of the fastest repetition" % folder)
    print("| forms | us/step | vs default | k_filter | k_edges1 | k_lap | prep | apply1 | stage2 | merge | dedupe | errors |\n|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, ds in sorted(rows.items(), key=lambda kv: min((d["ms_per_step"] for d in kv[1]), default=9e9)):
        if not ds:
            print("| %s | (no line) |" % name)
            continue
        best = min(ds, key=lambda d: d["ms_per_step"])
        s = best.get("segments_us", {})
        us = 1e3 * best["ms_per_step"]
        print("| %s | %.1f | %s | %s | %s |" % (name, us, ("%+.1f %%" % (100 * (us / b_us - 1))) if b_us else "",
                                              " | ".join("%.1f" % s.get(k, float("nan")) for k in ("k_filter", "k_edges1", "k_lap", "prep", "apply1", "stage2", "merge", "dedupe")),
                                              best.get("invalid", "")[:40]))
    print()


def main():
    args = sys.argv[1:]
    mats = []
    while "--matrix" in args:
        i = args.index("--matrix")
        mats.append(args[i + 1])
        del args[i:i + 2]
    if not args and not mats:
        args = [p for p in ("gpurun_out/r5a/bench_default.json", "gpurun_out/r5a/bench_driver.json") if os.path.exists(p)]
        mats = [p for p in ("gpurun_out/r5b",) if os.path.isdir(p)]
    for p in args:
        line_report(p)
    for m in mats:
        matrix_report(m)


if __name__ == "__main__":
    main()
