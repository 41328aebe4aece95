#!/usr/bin/env python
"""Tracker-only benchmark: BoT-SORT update steps / second on pre-extracted detections
(BASELINE.json metric), one independent video stream per GPU (no data-path collective).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|c1|c2|c3|c5|crowd|hard-reid]

A "step" is one whole BoTSORT.update (tracking half, reference D:2129-2133 / D:1414-1639) = one
`bts_tracker_step` on one frame of the seeded synthetic stream.  Frames are generated on the
host and uploaded to HBM before the timed region.  The default workload is the metric's headline size,
10 000 tracks x 2 000 detections x 2048-d.

`--gpus N` without a torch.distributed environment spawns N ranks itself (one process per GPU, one video
stream each, gloo control plane on 127.0.0.1: start/stop barriers and the max-over-ranks time -- no collective
on the data path); under `python -m torch.distributed.run` the environment's RANK / WORLD_SIZE are used.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline          .kernel: the dominant kernel of the step, timed with hipEvents on the step's stream in a
                    SEPARATE pass after the timed region (an event pair costs the stream ~4 us), against its
                    bound (bound type explicit); .step: the whole step against the 8 TB/s HBM roofline, with
                    SURVEY.md 8(d)'s algorithmic-bytes formula and the as-built variant
  export_inclusive  steps/s when every frame's result also leaves the device (MOT rows of the tracked list
                    written into a ring of pinned host buffers, bts_tracker_export_mot_async + one event per
                    frame, read one ring turn late), and with a host synchronisation per frame
  cpu_baseline      the CPU oracle (structure-faithful port of the reference: per-pair Python IoU, scalar Kalman
                    updates, dense extended-matrix JV) on the same stream, 1 core, bounded
  id_match          the metric's "assignment ID match vs ref": a second tracker handle stepped on the frames of
                    the oracle sample, compared frame by frame (assignment pairs of the three stages, ordered
                    track ids / states, Kalman means)
  facade            what a drop-in caller of BoTSORT.update() gets (D:2130-2185): steps/s of BoTSORT.update_arrays(export=True)
                    -- (a) the call alone (lazy track lists), (b) plus reading tlbr + track_id of EVERY returned track like the
                    reference's drawing loop, (c) with eager lists (round-2 behaviour) -- at this workload, and at c2 when
                    the workload is the headline size
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
I8_MFMA_PEAK_TF = 5000.0       # dense int8 MFMA peak (the same 5 P(FL)OP/s as fp8; not the 2:1-sparse headline): the pre-filter multiplies int8 shadows
F32_MFMA_PEAK_TF = 157.3       # dense f32-input MFMA peak (= f32 vector peak)
SEGMENTS = ("prep", "cos1", "cost1", "lap1", "apply1", "stage2", "merge", "dedupe")   # launch groups of a step
KERNELS = {   # single kernels that can dominate a step -> name in the rocprofv3 kernel trace
    "k_filter": "k_filter_i8<false> | k_filter_splitk<false> | k_filter_t<*>", "k_lap": "k_lap<4|8>", "k_edges1": "tk_edges<1,1,*> | tk_edges_tall<*> | tk_edges_tall_gate<*> | tk_edges_tall_gate_c<*>", "k_mid2": "tk_mid2<1>",
    "k_apply23": "tk_apply23", "prep": "tk_front", "merge": "tk_merge",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target")
    ap.add_argument("--segment", default="auto", help="kernel timed for roofline.kernel (auto = the dominant one)")
    ap.add_argument("--roofline-steps", type=int, default=48, help="steps of the separate event pass behind the timed region")
    ap.add_argument("--streams-per-gpu", type=int, default=1, help="independent video streams per GPU, each on its own HIP stream and "
                    "host thread (the headline metric is 1; more shows what a GPU serves when steps are launch-bound)")
    ap.add_argument("--several-streams", action="store_true", help="also report 4 and 8 concurrent streams on this GPU (fresh processes)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline, the ID match and the standalone kernels")
    ap.add_argument("--no-export", action="store_true", help="skip the export-inclusive legs")
    ap.add_argument("--id-frames", type=int, default=0, help="frames of every rank's ID match under --gpus N (default: 24, or 4 for big tables)")
    ap.add_argument("--no-standalone", action="store_true", help="skip the standalone kernel rooflines of the small workloads")
    ap.add_argument("--dense", action="store_true", help="dense float32 appearance path instead of the int8 pre-filter")
    ap.add_argument("--graph", action="store_true", help="replay each step as one hipGraph (bts_tracker_config.use_graph)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-segment timing table to stderr")
    ap.add_argument("--motion-gate", action="store_true", help="Mahalanobis gate on the stage-1 pairs (bts_tracker_config.motion_gate; "
                    "off = the reference's cascade).  BASELINE config 3 is stated with the gate on: --workload c3 implies it")
    ap.add_argument("--no-motion-gate", action="store_true", help="c3 without the gate")
    ap.add_argument("--forms", default="auto", help="execution forms of the tracker (bts_tracker_create_forms; they never change a result): "
                    "auto = selected on this GPU in the untimed set-up (tune.run_isolated: a child process steps each candidate on frames "
                    "of this workload, keeps what is bit-identical to the default forms AND faster; the line carries the whole table), "
                    "default = the library's defaults, or an explicit 'KEY=VALUE,...'")
    ap.add_argument("--compare-forms", default="", help="'KEY=VALUE,...': no selection -- the default forms and THESE forms through the same timed region on "
                    "the same frames, both figures (and both per-kernel tables) in the line; `value` stays the default forms' figure")
    ap.add_argument("--oversubscribe", action="store_true", help="allow --gpus N on a box with fewer GPUs (ranks share devices; "
                    "for testing the launch path only -- the line says so)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- launch
def self_spawn(a) -> int:
    """`--gpus N` without a distributed environment: N ranks of this script, one per GPU (reference deployment:
    one process per video stream, D:390 / D:1264).  Rank 0's stdout (the JSON line) is passed through."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus and not a.oversubscribe:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible (use --oversubscribe to share devices for a launch-path test)\n"
                         % (a.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BTS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    out = "".join(ln + "\n" for ln in out.splitlines() if ln.startswith("{"))        # the JSON line, nothing else
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


# ------------------------------------------------------------------------------------------- models
def kernel_model(kern, K, M, D, dense, filt=None):
    """Algorithmic work of ONE launch of a kernel in a steady-state frame (pool N = K rows, M detections,
    7/8 of them high-score, ~M matches, ~1.2 feasible edges per detection): (bound, amount, unit, peak).
    Formulas: DESIGN.md section 3."""
    N, Mh, Km = K, M - M // 8, M - M // 8
    E = int(1.2 * Mh)
    if kern == "k_filter":
        # appearance contraction on the int8 matrix cores: the multiply-adds the kernel EXECUTED (tiles stop at a prefix of K
        # once nothing in them can reach the gate: counted on the device, 128 x 128 x 128 per tile stage), not 2 N Mh D
        if filt and filt.get("k_stages"):
            tm, tn = filt.get("tile") or (128, 128)       # (BTS_FILTER_TILE: a K stage of the one-pass form is tm x tn x 128)
            return "mfma", 2.0 * tm * tn * 128 * filt["k_stages"], "op", I8_MFMA_PEAK_TF
        return "mfma", 2.0 * N * Mh * D, "op", I8_MFMA_PEAK_TF
    table = {
        # stage-1 matches: Kalman rows read + written, measurement read; det row, smooth r+w, curr w, int8 shadow w
        "k_mid2": Km * (2 * 72 + 4) * 8 + Km * D * (4 + 4 + 4 + 4 + 2),
        "k_apply23": (M // 8) * ((2 * 72 + 4) * 8 + D * 18),    # the low-score matches of stage 2 (stage 3 is empty in steady state)
        "k_edges1": (N + Mh) * 32 + E * 16 + Mh * 2 * D * 4,    # boxes in, sparse edges out, two feature rows per refined candidate
        "k_lap": E * 16 + 8 * Mh,                               # edge list in, assignment vectors out
        "prep": M * D * 6 + M * 48,                             # detection rows read (f32) + int8 shadows written, boxes
        "merge": (N + M) * 16,                                  # list entries and per-slot flags
    }
    return "hbm", float(table[kern]), "byte", HBM_PEAK_GBS


def step_bytes(K, M, D):
    """Algorithmic HBM bytes of ONE step, two ways (both stated in DESIGN.md section 3):
    survey   SURVEY.md 8(d)'s formula with its fp32 element sizes (s_k = s_f = s_c = 4), K_matches = M:
             2 N 72 s_k + K (2 72 + 4) s_k + (N + M) 16 + (N + M) D s_f + 2 N M s_c + 4 K D s_f + 4 (N + M)
    as_built what this implementation has to move: float64 Kalman rows (s_k = 8), no N x M matrix at all (the 2 N M s_c
             term is gone), each track / high-score detection feature row read once as its 1-byte int8 shadow by the
             pre-filter, detection rows read once in float32 and their shadows written, exact float32 refinement of one
             candidate per high-score detection (two rows), feature write-back 18 D bytes per match."""
    N, Km, Mh = K, M, M - M // 8
    survey = 2 * N * 72 * 4 + Km * (2 * 72 + 4) * 4 + (N + M) * 16 + (N + M) * D * 4 + 2 * N * M * 4 + 4 * Km * D * 4 + 4 * (N + M)
    as_built = (2 * N * 72 * 8 + Km * (2 * 72 + 4) * 8 + (N + M) * 48 + (N + Mh) * D * 1 + M * D * (4 + 1)
                + Mh * 2 * D * 4 + Km * D * 18 + 16 * int(1.2 * Mh) + 4 * (N + M) * 3)
    return float(survey), float(as_built)


def table_rows(K, motion_gate=False):
    """Track-table capacity for a stream of K identities: the live tracks plus room for the duplicates a long run
    accumulates (with the motion gate a re-appearing identity whose prediction drifted is born again while its old
    track waits out the 300-frame lost buffer, so a few hundred frames pile up some 0.15 K of them; the reference's
    lists are unbounded, a table that fills up is reported as a device error and voids the run)."""
    return K + (K // 2 if motion_gate else max(64, K // 8))


class IdMatch:
    """Checker beside the CPU-baseline leg (the metric's "assignment ID match vs ref"): a second tracker handle is
    stepped on the very frames the oracle is timed on (outside the oracle's timed bracket) and, per frame, the three
    stages' assignment indices, the ordered track ids / states of the tracked and lost lists and the Kalman means are
    compared with the oracle's."""
    INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")

    def __init__(self, pkg, K, M, D, motion_gate=False, forms=None, **opts):
        import torch
        self.torch = torch
        # (the checker runs the SAME execution forms as the timed tracker: what is measured is what is checked)
        self.trk = pkg.DeviceTracker(max_tracks=table_rows(K, motion_gate), max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(),
                                     motion_gate=1 if motion_gate else 0, forms=forms or None, **opts)
        self.frames = self.pairs = self.pairs_equal = self.ids = self.ids_equal = 0
        self.frames_identical = 0
        self.kalman_err = 0.0

    def frame(self, fr, feats, ora, table):
        t = self.torch
        self.trk.step(t.from_numpy(fr.boxes).cuda(), t.from_numpy(fr.scores).cuda(), t.from_numpy(feats).cuda())
        same = True
        for k in (1, 2, 3):
            got, want = self.trk.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)
            got = np.asarray(got).reshape(-1, 2)
            eq = int((got == want).all(axis=1).sum()) if got.shape == want.shape else len(set(map(tuple, got)) & set(map(tuple, want)))
            self.pairs += len(want)
            self.pairs_equal += eq
            same &= got.shape == want.shape and eq == len(want)
        for which, lst in ((0, ora.tracked), (1, ora.lost)):
            g, w = self.trk.export(which), table(lst)
            ok = all(np.array_equal(g[key], w[key]) for key in self.INT_KEYS)
            self.ids += len(w["track_id"])
            self.ids_equal += len(w["track_id"]) if ok else int(np.sum(g["track_id"][:len(w["track_id"])] == w["track_id"][:len(g["track_id"])]))
            same &= ok
            if ok and len(w["track_id"]):
                self.kalman_err = max(self.kalman_err, float(np.abs(g["mean"] - w["mean"]).max()))
        self.frames += 1
        self.frames_identical += int(same)

    def result(self):
        errors = self.trk.counts()["errors"]
        self.trk.close()
        return {"frames": self.frames, "frames_identical": self.frames_identical,
                "assignment_pairs": self.pairs, "assignment_pairs_equal": self.pairs_equal,
                "track_ids": self.ids, "track_ids_equal": self.ids_equal,
                "kalman_mean_max_abs_err": self.kalman_err, "kalman_tolerance": 1e-4, "device_errors": errors,
                "against": "CPU oracle (pinned to the reference's golden vectors), every frame of the sample incl. "
                           "the two pool-building frames; tracked + lost lists in order, stages 1-3 (row, column) pairs"}


def cpu_model():
    """The host CPU's model string (the CPU figures are quoted with core count and model)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(pkg, wl, seed, budget_s, checker=None, faithful=True, max_frames=200, motion_gate=False):
    """The oracle timed on this box's host cores: 1 process, 1 core (the reference is
    single-threaded Python).  Only this leg may touch oracle/."""
    from oracle import bts_oracle as bo
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    st = pkg.synth.make_workload(wl, seed=seed)
    ora = bo.TrackerOracle(faithful_iou=faithful, motion_gate=motion_gate)
    t_used, n = 0.0, 0
    f = 0
    while True:
        fr = st.next()
        feats0 = fr.feats.copy() if checker is not None else None     # (the oracle, like the reference, renormalises rows of its input in place: Q8)
        t0 = time.perf_counter()
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats, None)
        dt = time.perf_counter() - t0
        if checker is not None:
            checker.frame(fr, feats0, ora, bo.TrackerOracle.table)
        f += 1
        if f <= 2:          # frames 0-1 build the track pool: not steady state
            if dt > budget_s * 4:
                break
            continue
        t_used += dt
        n += 1
        if t_used >= budget_s or n >= max_frames:
            break
    if limit is not None:
        limit.unregister() if hasattr(limit, "unregister") else None
    if n == 0:
        return {"value": None, "unit": "steps/s", "cores": 1, "kind": "port",
                "sample": "not timed (--cpu-seconds 0)" if budget_s <= 0 else "frame 1 alone exceeded the time budget at this size"}
    form = ("structure-faithful oracle (per-pair Python IoU, scalar Kalman update, dense JV on the (N+M)^2 matrix)" if faithful else
            "vectorised NumPy form of the same oracle (matrix IoU, batched Kalman; identical results)")
    return {"value": n / t_used, "unit": "steps/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count() or 0,
            "sample": "%d steady-state frames (frames 2..%d) of the same seeded stream, %.1f s, %s; host has %d cores"
                      % (n, n + 1, t_used, form, os.cpu_count() or 0)}


def pmc_traffic(workload, kernel_name, lib_path=None):
    """(HBM bytes per launch of the roofline kernel, provenance) from the committed rocprofv3 --pmc passes (FETCH_SIZE /
    WRITE_SIZE, separate runs: scripts/gpu_pmc_traffic.sh -> scripts/pmc_to_json.py).  The counters belong to the machine
    code they were taken on: every entry of profiles/pmc_traffic.json carries the hash of its kernel's instruction stream
    (scripts/kernel_hashes.py --stamp), and the figure is quoted only when the LOADED library holds a kernel of that name
    with the same hash -- otherwise traffic is None and the provenance says so."""
    src = {"file": "profiles/pmc_traffic.json", "taken_at_commit": None, "kernel_hash": None, "loaded_library_matches": None}
    try:
        doc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        src["taken_at_commit"] = doc.get("taken_at_commit")
        tab = doc["workloads"][workload]
    except Exception:
        src["note"] = "no counter pass covers this workload"
        return None, src
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    try:
        import kernel_hashes as kh
        live = kh.library_hashes(lib_path) if lib_path else {}
    except Exception as e:
        src["note"] = "kernel hashes of the loaded library unavailable (%s): counters not quoted" % str(e)[:120]
        return None, src
    bases = [alt.strip().split("<")[0] for alt in kernel_name.split("|")]      # (a kernel may have several forms)
    live_hashes = {h for n, h in live.items() if kh.base(n) in bases}
    best, stale = None, False
    for k, v in tab.items():
        if k.split("<")[0] not in bases or v.get("traffic_bytes") is None:
            continue
        if v.get("kernel_hash") and v["kernel_hash"] in live_hashes:
            if best is None or v["traffic_bytes"] > best:      # the form that ran (the others are absent or tiny)
                best, src["kernel_hash"] = v["traffic_bytes"], v["kernel_hash"]
        else:
            stale = True
    src["loaded_library_matches"] = best is not None
    if best is None and stale:
        src["note"] = "the kernel's machine code has changed since the counter pass: re-run scripts/gpu_pmc_traffic.sh"
    return best, src


def several_streams(a, counts):
    """Aggregate steps/s of n concurrent streams on this GPU, n in `counts`: `bench.py --streams-per-gpu n`, each in a
    fresh process (the runtime binds HIP streams to its hardware queues at first use; a process that has already used
    other streams gets less regular bindings)."""
    out = []
    for n in counts:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", a.workload, "--steps", str(min(a.steps, 200)),
               "--warmup", str(a.warmup), "--no-cpu", "--no-export", "--streams-per-gpu", str(n)] + (["--dense"] if a.dense else []) + (["--motion-gate"] if a.motion_gate else ["--no-motion-gate"])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out.append({"streams": n, "value": d["value"], "unit": d["unit"], "steps_per_stream": d["steps"]})
        except Exception as e:          # a reported extra, never fatal for the headline line
            out.append({"streams": n, "value": None, "error": str(e)[:200]})
    return out


T_START = time.time()


def trace_legs(workload):
    """The trace VARIANTS of the library (ab/ftrace.so, ab/trace.so: scripts/prepare_gpu_batches.sh; git-ignored, they travel with the
    working tree) decoded by their scripts in child processes: k_filter_i8's own cycle accounting -- the decomposition of the
    pre-filter's "fixed 22.5 us" (scripts/dbg_filtertrace.py) -- and k_lap's phase timeline (scripts/dbg_laptrace.py).  Last legs of
    the line, own processes, own library: nothing measured above is touched.  They are in the line so that the ONE bench run a
    round may get on a GPU also answers the questions the next round's work depends on."""
    out = {}
    if time.time() - T_START > 480:          # (extras never push the line past the driver's patience: skipped when the run is already long)
        return {"skipped": "the run had taken %d s when the trace legs were due" % int(time.time() - T_START)}
    for key, lib, script in (("prefilter_phases", "ftrace.so", "dbg_filtertrace.py"), ("lap_phases", "trace.so", "dbg_laptrace.py"),
                             ("list_job_phases", "trace.so", "dbg_jobtrace.py")):
        path = os.path.join(REPO, "ab", lib)
        if not os.path.exists(path):
            continue
        try:
            env = dict(os.environ, BTS_LIB_PATH=path)
            for k in [k for k in env if k.startswith("BTS_") and k != "BTS_LIB_PATH"]:
                env.pop(k)
            r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", script), workload], cwd=REPO, env=env, capture_output=True, text=True, timeout=100)
            out[key] = {"library": "ab/" + lib, "rc": r.returncode, "table": [ln for ln in r.stdout.splitlines() if ln.strip()][-30:],
                        "stderr_tail": r.stderr[-300:] if r.returncode else ""}
        except Exception as e:
            out[key] = {"library": "ab/" + lib, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def guarded(name, fn, *args, **kw):
    """Every leg BEHIND the timed region is an extra: whatever it raises becomes {"error": ...} in its object and the
    headline line (value, roofline, cpu_baseline) is still printed -- a first-contact failure in the export, facade or
    checker code must not cost the measurement."""
    try:
        return fn(*args, **kw)
    except BaseException as e:          # (incl. SystemExit / assertion failures of a leg)
        if isinstance(e, KeyboardInterrupt):
            raise
        import traceback
        sys.stderr.write("bench.py: leg %r failed:\n%s\n" % (name, traceback.format_exc()))
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def ms_seed(pkg, rank):
    return pkg.multistream.stream_seed(1234, rank)      # one independent video stream per GPU


def export_legs(pkg, trk, frames, torch):
    """What a per-frame consumer (D:2135-2185) sees.  (a) asynchronous: every step is followed by
    bts_tracker_export_mot_async into the next buffer of a ring of pinned host buffers and an event; the host reads a
    buffer one ring turn later (waits on that frame's event only).  (b) synchronous: bts_tracker_export_mot (stream
    synchronisation + copy) after every step."""
    RING = 4
    T = trk.cfg.max_tracks
    ring = [pkg.PinnedMot(T) for _ in range(RING)]
    events = [torch.cuda.Event() for _ in range(RING)]
    used = [False] * RING
    rows_seen = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, fr in enumerate(frames):
        slot = k % RING
        if used[slot]:
            events[slot].synchronize()
            rows_seen += ring[slot].count()                      # the consumer's read of that frame
        trk.step(*fr)
        trk.export_mot_async(ring[slot])
        events[slot].record()
        used[slot] = True
    for slot in range(RING):
        if used[slot]:
            events[slot].synchronize()
            rows_seen += ring[slot].count()
    t1 = time.perf_counter()
    n_async = len(frames)
    dt_async = t1 - t0
    for r in ring:
        r.close()
    half = frames[: max(8, len(frames) // 4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows_sync = 0
    for fr in half:
        trk.step(*fr)
        rows_sync += len(trk.export_mot())
    t1 = time.perf_counter()
    return {"async_steps_per_s": n_async / dt_async, "async_us_per_step": 1e6 * dt_async / n_async,
            "async_mean_rows_per_frame": rows_seen / max(n_async, 1), "ring": RING,
            "sync_steps_per_s": len(half) / (t1 - t0), "sync_us_per_step": 1e6 * (t1 - t0) / len(half),
            "sync_frames": len(half), "sync_mean_rows_per_frame": rows_sync / len(half),
            "what": "async: bts_tracker_export_mot_async (MOT rows written by the export kernel into pinned host memory, one event "
                    "per frame, read one ring turn late); sync: bts_tracker_export_mot (stream sync + copy) after every step"}


def facade_leg(pkg, torch, wl, n_frames, seed, forms=None):
    """The BoTSORT facade's own rate on one GPU: update_arrays(export=True) fed with frames already resident in HBM.
    lists_only: the call returns lazy TrackLists (one bts_tracker_export_lists per frame, no per-track Python);
    read_all:   the caller then reads tlbr and track_id of every returned track (the reference's loop D:2135-2143);
    eager:      lazy_lists=False -- every STrack mirror built per frame (what round 2 did unconditionally)."""
    K, M, D = pkg.synth.workload_dims(wl)
    st = pkg.synth.make_workload(wl, seed=seed)
    warm = 4

    def frames(n):
        out = []
        for _ in range(n):
            fr = st.next()
            out.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()))
        return out

    res = {"workload": wl, "frames_per_leg": n_frames}
    for name, lazy, read in (("lists_only", True, False), ("read_all", True, True), ("eager", False, False)):
        bot = pkg.BoTSORT(max_tracks=table_rows(K), max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(), lazy_lists=lazy, forms=forms or None)
        st = pkg.synth.make_workload(wl, seed=seed)
        for fr in frames(warm):
            bot.update_arrays(*fr)
        fl = frames(n_frames if name != "eager" else max(8, n_frames // 4))
        torch.cuda.synchronize()
        acc = 0
        t0 = time.perf_counter()
        for fr in fl:
            out = bot.update_arrays(*fr)
            if read:
                for t in out:
                    acc += t.track_id + int(t.tlbr[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name + "_steps_per_s"] = len(fl) / dt
        res[name + "_us_per_step"] = 1e6 * dt / len(fl)
        res["mean_tracked"] = len(bot.tracked_stracks)
        bot.device_tracker.close()
    res["what"] = ("BoTSORT.update_arrays(export=True) on device-resident frames; lists_only = the call (lazy TrackLists), read_all = + tlbr and "
                   "track_id of every returned track, eager = lazy_lists=False (every mirror built per frame)")
    return res


class LineGuard:
    """ONE JSON line on rank 0's stdout, whatever happens behind the first measurement.  `partial` holds the contract fields of the
    measurement(s) taken so far; if the legs behind them (form selection, event passes, export / facade / checker legs -- code that
    may be meeting a GPU for the first time) have not finished `seconds` after arm(), the watchdog prints `partial` with a
    "truncated" note and ends the process.  emit() prints the complete line instead; whichever comes first wins."""

    def __init__(self, rank):
        import threading
        self.rank, self.lock, self.done, self.partial, self.timer = rank, threading.Lock(), False, None, None

    def arm(self, partial, seconds):
        import threading
        self.partial = partial
        self.timer = threading.Timer(seconds, self._fire, args=(seconds,))
        self.timer.daemon = True
        self.timer.start()

    def _fire(self, seconds):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0 and self.partial is not None:
                line = dict(self.partial, truncated="the legs behind the timed region did not finish within %d s: this line carries the "
                                                    "measurements completed until then (watchdog, bench.py LineGuard)" % seconds)
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            os._exit(0 if self.rank == 0 else 3)

    def emit(self, out):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.timer is not None:
                self.timer.cancel()
            print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_spawn(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (gloo announces its connections on stdout; rank 0's stdout carries the ONE JSON line only)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required (no CPU fallback)"
    ndev = torch.cuda.device_count()
    if world > ndev and not a.oversubscribe:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, ndev))
    torch.cuda.set_device(local % ndev)
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    K, M, D = pkg.synth.workload_dims(a.workload)
    a.motion_gate = (a.motion_gate or a.workload == "c3") and not a.no_motion_gate
    a.warmup = max(a.warmup, 3)          # (frames 0-1 build the track pool, frame 2 is the first steady-state one: never timed)

    S = max(1, a.streams_per_gpu)
    st = pkg.synth.make_workload(a.workload, seed=ms_seed(pkg, rank * S))
    PROBE = 4

    def next_frame():
        fr = st.next()
        return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())

    stream = torch.cuda.Stream()          # graph replay needs a real (non-legacy) stream
    torch.cuda.set_stream(stream)
    opts = dict(appearance_mode=1 if a.dense else 0, use_graph=1 if a.graph else 0, motion_gate=1 if a.motion_gate else 0)
    FORM_KEYS = ("BTS_EDGES_SORTED", "BTS_EDGES_GRID", "BTS_LISTS_AHEAD", "BTS_MERGE_EARLY", "BTS_RENORM_FUSED", "BTS_CVT_OVERLAP", "BTS_PREDICT_FIRST",
                 "BTS_PREDICT_FRONT", "BTS_FILTER_GUESS", "BTS_LAP_FORM", "BTS_LAP_HELPERS", "BTS_CHUNK_ROWS", "BTS_FILTER_PW", "BTS_I8_FIRST16",
                 "BTS_TILE_GOAL", "BTS_FILTER_ONEPASS", "BTS_SHADOW_PLANES", "BTS_LAP_SELF", "BTS_FILTER_TILE", "BTS_FILTER_AHEAD", "BTS_FILTER_AHEAD_PW", "BTS_SMOOTH_DEFERRED")
    if a.forms == "auto" and any(k in os.environ for k in FORM_KEYS):
        a.forms = "env"                 # (forms given through the environment -- an A/B run -- are taken as they are: no selection)
    auto = a.forms == "auto" and S == 1 and not a.graph and not a.dense and not os.environ.get("BTS_BENCH_NO_TUNE") and not a.compare_forms
    forms0 = "" if a.forms in ("auto", "default", "env") else a.forms        # what the FIRST measurement runs: the defaults, or the caller's
    # every frame of the run, uploaded once and kept pristine: a measurement steps on its own copy of the feature rows (the step
    # renormalises used rows in place, Q8), so the default and the selected forms are timed on bit-identical inputs
    pristine = [next_frame() for _ in range(a.warmup + a.steps)]
    ms = importlib.import_module("bot-sort-onnx-tensorrt_amd.multistream")
    ddist = dist if world > 1 else None
    IGNORE = bool(os.environ.get("BTS_BENCH_IGNORE_ERRORS"))       # (kernel timing experiments whose results are meaningless)

    def measure(forms):
        """warm-up + the timed region for one tracker with `forms`: -> dict(trk, elapsed (max over ranks), counts, timed_errors,
        timed (its frames), extra).  K steps, nothing else on the stream (no events, no read-backs), barrier + synchronize on both
        sides (the contract of the task statement)."""
        # Every rank passes the SAME sequence of collectives (barrier, barrier, three all-reduces) whatever happens to its own leg:
        # a rank whose tracker cannot be created or whose step raises still meets the others at the barriers, and the failure is
        # then raised on every rank (a guarded second measurement that failed on one rank left the others waiting).
        fail, trk, timed, extra, t0, t1 = None, None, [], [], 0.0, 0.0
        try:
            mine = [(b, sc, f.clone()) for (b, sc, f) in pristine]
            torch.cuda.synchronize()
            frames, timed = mine[:a.warmup], mine[a.warmup:]
            o = dict(opts, forms=forms or None)
            trk = pkg.DeviceTracker(max_tracks=table_rows(K, a.motion_gate), max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(), **o)
            if hasattr(pkg._lib.load(), "bts_tracker_inputs_resident"):
                trk.inputs_resident(True)    # (every frame is uploaded and synchronised before the loop that steps on it; only the
                                             #  opt-in BTS_CVT_OVERLAP pipelining looks at this promise)
            for k in range(a.warmup):
                trk.step(*frames[k])
            torch.cuda.synchronize()
            extra = []                       # further streams of this GPU: (tracker, HIP stream, frames), warmed up the same way
            for k in range(1, S):
                st_k = pkg.synth.make_workload(a.workload, seed=ms_seed(pkg, rank * S + k))
                with torch.cuda.stream(torch.cuda.default_stream()):      # (set-up on the default stream: see multistream.run_concurrent)
                    tk = pkg.DeviceTracker(max_tracks=table_rows(K, a.motion_gate), max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(), **o)
                    for _ in range(a.warmup + 6):
                        fr = st_k.next()
                        tk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
                    fl = []
                    for _ in range(a.steps):
                        fr = st_k.next()
                        fl.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()))
                extra.append((tk, None, fl))
            if a.graph:                                              # capture happens on the first replayed step: keep it untimed
                trk.step(*next_frame())
                torch.cuda.synchronize()
        except BaseException as e:      # noqa: BLE001  (re-raised below, behind the collectives)
            fail = e
        # ---- the timed region
        ms.barrier(ddist)
        if fail is None:
            try:
                torch.cuda.synchronize()
                if S == 1:
                    t0 = time.perf_counter()
                    for fr in timed:
                        trk.step(*fr)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                else:
                    t0 = 0.0
                    t1 = ms.run_concurrent([trk] + [e[0] for e in extra], [timed] + [e[2] for e in extra])
            except BaseException as e:  # noqa: BLE001
                fail = e
        ms.barrier(ddist)
        any_fail = ms.max_over_ranks(1.0 if fail is not None else 0.0, ddist) > 0
        if any_fail:
            ms.max_over_ranks(0.0, ddist); ms.max_over_ranks(0.0, ddist)      # (the two all-reduces of the success path)
            if fail is not None:
                raise fail
            raise RuntimeError("the measurement failed on another rank")
        counts = trk.counts()
        # a device-side error in the timed region (a truncated frame: table full, edge list overflow, an in-kernel wait that timed
        # out) does not stop the line from being printed -- it is printed WITH the verdict that the measurement is invalid
        errs = 0 if IGNORE else int(counts["errors"]) + sum(int(e[0].counts()["errors"]) for e in extra)
        errs = int(ms.max_over_ranks(float(errs), ddist))         # (every rank takes the same decisions below)
        for e in extra:
            e[0].close()
        return dict(trk=trk, elapsed=ms.max_over_ranks(t1 - t0, ddist), counts=counts, timed_errors=errs, timed=timed, forms=forms)

    # ---- 1. the DEFAULT forms (or the caller's) first: this measurement exists before any never-run candidate form is stepped
    #         (a wedged candidate then costs the extras, not the line)
    try:
        first = measure(forms0)
    except BaseException as e:
        # the product path failed outright (no library, a HIP error at create or in a step): there is no number, and the line says so
        # instead of leaving stdout empty -- value null, the error in plain text, non-zero exit
        if rank == 0 and not isinstance(e, KeyboardInterrupt):
            print(json.dumps({"metric": "tracker update-steps/sec (N_tracks x M_dets)", "value": None, "unit": "steps/s", "n_gpus": world,
                              "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "config": {"workload": a.workload}, "roofline": None, "cpu_baseline": None,
                              "error": "the timed region of the default forms did not complete: %s: %s" % (type(e).__name__, str(e)[:400])}), flush=True)
        raise
    run = first
    selection, second, any_selected = None, None, False

    def contract(elapsed_s, which):
        """the contract fields of one completed measurement (nothing here touches the device)"""
        return {"metric": "tracker update-steps/sec (N_tracks x M_dets)", "value": ms.aggregate_throughput(a.steps * S, elapsed_s, world),
                "unit": "steps/s", "value_is": which, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed_s / a.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic (seeded stream generator, SURVEY.md 8d; frames resident in HBM before the timed region)",
                "config": {"workload": "%s: %s" % (a.workload, pkg.synth.workload_text(a.workload)), "tracks": K, "dets": M, "feat_dim": D,
                           "streams": world * S, "forms": {"mode": a.forms, "requested": forms0}},
                "roofline": None, "cpu_baseline": None}
    guard = LineGuard(rank)
    partial = contract(first["elapsed"], "default forms" if forms0 == "" else "requested forms")
    if first["timed_errors"]:
        partial["invalid"] = "device error flags %d raised inside the timed region" % first["timed_errors"]
    guard.arm(partial, float(os.environ.get("BTS_BENCH_WATCHDOG_S", "780")))
    if auto:
        # ---- 2. execution forms selected ON this GPU, in a child process (tune.run_isolated: a candidate that takes the child down
        #         or hangs it is blamed and skipped; nothing of it runs in this process)
        tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
        selection = guarded("form_selection", tune.run_isolated, a.workload, ms_seed(pkg, rank * S), local % ndev,
                            n_frames=int(os.environ.get("BTS_TUNE_FRAMES", "116")), timeout_s=float(os.environ.get("BTS_TUNE_TIMEOUT", "300")),
                            repeats=int(os.environ.get("BTS_TUNE_REPEATS", "3")),
                            tracker_args=dict(max_tracks=table_rows(K, a.motion_gate), max_dets=max(K, M), feat_dim=D, **opts))
        sel_forms = selection.get("forms") if isinstance(selection, dict) and "error" not in selection else ""
        # ---- 3. the selected forms through the SAME timed region (same frames, same barriers); under --gpus N every rank
        #         re-measures (with its own selection, possibly the defaults again) so that the barriers pair up
        any_selected = ms.max_over_ranks(1.0 if sel_forms else 0.0, ddist) > 0      # (rank-agreed)
        if world > 1 or sel_forms:
            second = guarded("selected_forms", measure, sel_forms or "")
            if isinstance(second, dict) and "error" in second:
                second = None
        if second is not None:
            # WHICH figure becomes `value` is the selection's verdict (3 x 110 steps per candidate, the default measured twice as the
            # noise floor, 2 % margin), not a comparison of this --steps-long re-run with the first one: the better of two short
            # timed regions is biased upward by their run-to-run spread.  Forms were selected (on some rank)
            # -> the re-run with them is `value`; nothing selected -> the default forms' first measurement is, and under --gpus N the
            # second run of the same forms is only reported (value_selected_forms), never taken as a maximum.
            ok2 = second["timed_errors"] == 0 or first["timed_errors"] != 0
            if ok2 and any_selected:
                run = second
    if a.compare_forms and S == 1:
        # an A/B asked for by the caller: `value` stays the default forms' figure, the compared set is value_selected_forms
        second = guarded("compare_forms", measure, a.compare_forms)
        if isinstance(second, dict) and "error" in second:
            second = None
    loser = second if run is first else first
    other_segments = None
    if loser is not None and loser is not run:
        # the per-kernel picture of the forms that did NOT become `value` as well (same event passes as below, on frames behind the
        # timed ones): the line then holds a per-kernel A/B of the default forms against the selected ones
        def probe_other():
            seg_us = {}
            fr4 = [(b, sc, f.clone()) for (b, sc, f) in pristine[-PROBE:]]
            for seg in SEGMENTS + tuple(k for k in KERNELS if k not in SEGMENTS) + ("step",):
                loser["trk"].profile_enable(seg, PROBE)
                for fr in fr4:
                    loser["trk"].step(*fr)
                ms_, n_ = loser["trk"].profile_read()
                seg_us[seg] = round(1e3 * ms_ / max(n_, 1), 2)
            loser["trk"].profile_enable("step", 0)
            return seg_us
        if rank == 0 and second is not None and (second["forms"] or "") != (first["forms"] or ""):
            other_segments = guarded("segments_other_forms", probe_other)
        guarded("close", loser["trk"].close)
    trk, elapsed, counts, timed_errors, timed, forms = run["trk"], run["elapsed"], run["counts"], run["timed_errors"], run["timed"], run["forms"]
    if run is second:
        better = contract(second["elapsed"], "selected forms" if (second["forms"] or any_selected) else "default forms")
        better["config"]["forms"]["requested"] = second["forms"]
        better["value_default_forms"], better["value_selected_forms"] = partial["value"], better["value"]
        if second["timed_errors"]:
            better["invalid"] = "device error flags %d raised inside the timed region" % second["timed_errors"]
        partial.clear()
        partial.update(better)

    # ---- behind the timed region: which kernel dominates a step, and its duration (separate event passes)
    breakdown = {}
    more = [next_frame() for _ in range(PROBE)]

    def probe_kernels():
        for seg in SEGMENTS + tuple(k for k in KERNELS if k not in SEGMENTS) + ("step",):
            trk.profile_enable(seg, PROBE)
            for fr in more:
                trk.step(*fr)
            ms_, n_ = trk.profile_read()
            breakdown[seg] = 1e3 * ms_ / max(n_, 1)
        segment = a.segment if a.segment != "auto" else max(KERNELS, key=lambda k: breakdown[k])
        roof_frames = (timed + more)[: a.roofline_steps] if len(timed) + len(more) >= a.roofline_steps else timed + more
        trk.profile_enable(segment, len(roof_frames))
        for fr in roof_frames:
            trk.step(*fr)
        torch.cuda.synchronize()
        seg_ms, seg_n = trk.profile_read()
        trk.profile_enable("step", 0)
        return segment, seg_ms, seg_n

    probe = guarded("kernel_probe", probe_kernels)
    probe_error = probe.get("error") if isinstance(probe, dict) else None
    segment, seg_ms, seg_n = ("k_filter", 0.0, 0) if probe_error else probe
    late_errors = guarded("device_errors", lambda: trk.counts()["errors"])
    if late_errors != 0 and not IGNORE and not isinstance(late_errors, dict):
        guarded("device_error_message", trk.export, 0)            # (logs the device's own account of what overflowed, and when)
    if a.breakdown and rank == 0:
        sys.stderr.write("segment us/step: " + json.dumps({k: round(v, 1) for k, v in breakdown.items()}) + "\n")

    export = None
    if rank == 0 and S == 1 and not a.no_export and not a.graph:
        export = guarded("export_inclusive", export_legs, pkg, trk, timed, torch)

    facade = None
    if rank == 0 and world == 1 and S == 1 and not a.no_export and not a.graph and not a.dense and not a.motion_gate:
        nf = max(16, min(a.steps, 64 if K <= 1024 else 24))      # (big workloads: generating a frame on the host takes ~0.4 s)
        facade = guarded("facade", facade_leg, pkg, torch, a.workload, nf, ms_seed(pkg, 0), forms)
        if a.workload == "target":
            facade = {"target": facade, "c2": guarded("facade_c2", facade_leg, pkg, torch, "c2", 4 * nf, ms_seed(pkg, 0))}

    big = K > 1024
    id_multi = None
    if world > 1 and not a.no_cpu and S == 1 and not a.dense:
        # ID match of EVERY rank's stream (N > 1: no CPU baseline, that is the N = 1 line's): steady-state frames of this
        # rank's stream through a second handle and the oracle (vectorised IoU form: the checker, not a timing), gathered
        def rank_ids():
            chk = IdMatch(pkg, K, M, D, a.motion_gate, forms=forms)
            cpu_baseline(pkg, a.workload, ms_seed(pkg, rank), 120.0, chk, faithful=False, max_frames=a.id_frames or (4 if big else 24), motion_gate=a.motion_gate)
            return dict(chk.result(), stream_seed=int(ms_seed(pkg, rank)))
        id_multi = [None] * world
        dist.all_gather_object(id_multi, guarded("id_match(rank %d)" % rank, rank_ids))      # (every rank takes part, whatever its checker did)

    if rank == 0:
        filt = guarded("prefilter", trk.debug_filter)
        filt_ok = isinstance(filt, dict) and "error" not in filt
        bound, amount, unit, peak = kernel_model(segment, K, M, D, a.dense, filt if filt_ok else None)
        seg_s = (seg_ms / 1e3) / max(seg_n, 1) if seg_n else None
        if bound == "mfma":
            achieved, runit = (amount / seg_s / 1e12 if seg_s else None), "TOP/s"
        else:
            achieved, runit = (amount / seg_s / 1e9 if seg_s else None), "GB/s"
        step_s = elapsed / a.steps
        b_survey, b_built = step_bytes(K, M, D)
        traffic = guarded("roofline.traffic", pmc_traffic, a.workload, KERNELS[segment], getattr(pkg._lib, "LIB_PATH", None))
        traffic, traffic_src = traffic if isinstance(traffic, tuple) else (None, traffic)
        kern_roof = {"kernel": KERNELS[segment], "bound": bound,
                     "bound_peak": "dense int8 MFMA 5 POP/s" if bound == "mfma" else "HBM3E 8 TB/s",
                     "achieved": achieved, "peak": peak, "unit": runit, "frac": achieved / peak if achieved is not None else None,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_unit": "bytes per launch (HBM, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, profiles/pmc_traffic.json)",
                     "avg_us": 1e6 * seg_s if seg_s else None, "launches_timed": seg_n,
                     "timed_in": "separate pass of %d steps behind the timed region, one hipEvent pair per launch on the step's stream" % seg_n,
                     "algorithmic_per_launch": amount, "algorithmic_unit": unit}
        if probe_error:
            kern_roof["error"] = probe_error
        out = {
            "metric": "tracker update-steps/sec (N_tracks x M_dets)",
            "value": ms.aggregate_throughput(a.steps * S, elapsed, world),
            "unit": "steps/s",
            # both measurements of the same timed region (same frames, same barriers): the library's default forms, taken FIRST, and
            # the forms selected on this GPU afterwards; `value` is the selected forms' figure when the selection (its own 3 x 110-step
            # statistics) kept any, else the default forms' -- never the better of the two short regions -- and the id_match /
            # roofline / export legs below run the forms of `value`
            "value_default_forms": ms.aggregate_throughput(a.steps * S, first["elapsed"], world) if forms0 == "" else None,
            "value_selected_forms": ms.aggregate_throughput(a.steps * S, second["elapsed"], world) if second is not None else None,
            "value_is": ("selected forms" if (second is not None and run is second and (second["forms"] or any_selected)) else
                         "default forms" if forms0 == "" else "requested forms"),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,      # (one step of every stream)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded stream generator, SURVEY.md 8d; frames resident in HBM before the timed region)",
            "config": {"workload": "%s: %s" % (a.workload, pkg.synth.workload_text(a.workload)) + (", Mahalanobis gate on" if a.motion_gate else "")
                                   + ", IoU+cosine fused cost, 3-stage association, one stream per GPU",
                       "tracks": K, "dets": M, "feat_dim": D, "streams": world * S, "parallelism": "%d stream%s/GPU, no collective" % (S, "" if S == 1 else "s (own HIP stream + host thread each)"),
                       # (BTS_FILTER_AHEAD is a THROUGHPUT mode -- frame f's pre-filter runs on a side stream under step f-1 -- and is
                       #  named here whenever it is in effect: the figure then needs frames stepped back to back)
                       "launch": ("hipGraph replay" if a.graph else "eager") + (", pre-filter ahead on a side stream (BTS_FILTER_AHEAD)"
                                                                                 if (guarded("forms", trk.forms) or {}).get("filter_ahead") == "1" else "")
                                                                              + (", smooth EMA behind the step on a third stream (BTS_SMOOTH_DEFERRED)"
                                                                                 if (guarded("forms", trk.forms) or {}).get("smooth_deferred") == "1" else ""),
                       "motion_gate": bool(a.motion_gate),
                       "spawn": "self-spawned ranks" if os.environ.get("BTS_BENCH_SPAWNED") else ("torch.distributed.run" if world > 1 else "single process"),
                       "oversubscribed": bool(world > ndev),
                       "opt_in": {k: v for k, v in os.environ.items() if k.startswith("BTS_") and k not in ("BTS_BENCH_SPAWNED", "BTS_LIB_PATH")},
                       # execution forms of the timed tracker (and of the id_match / facade trackers): what was asked for, what is
                       # in effect (bts_tracker_forms).  They never change a result; "auto" = selected on this GPU, table below
                       "forms": {"mode": a.forms, "requested": forms, "in_effect": guarded("forms", trk.forms)}},
            # flat fields = the dominant kernel (contract); .step is the whole step
            "roofline": kern_roof,
        }
        out["roofline"]["step"] = {
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "us_per_step": 1e6 * step_s,
            "algorithmic_bytes_survey_formula": b_survey, "achieved_survey": b_survey / step_s / 1e9, "frac_survey": b_survey / step_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_as_built": b_built, "achieved_as_built": b_built / step_s / 1e9, "frac_as_built": b_built / step_s / 1e9 / HBM_PEAK_GBS,
            "formula": "SURVEY.md 8(d) with fp32 element sizes (survey) / float64 Kalman rows, no N x M matrix, int8 shadows (as built): bench.py step_bytes()"}
        if timed_errors:
            out["invalid"] = ("device error flags %d raised inside the timed region: frames were truncated, `value` is NOT a measurement "
                              "of the full step (device message: %s)" % (timed_errors, str(guarded("device_error_message", trk.export, 0))[:300]))
        out["segments_us"] = {k: round(v, 2) for k, v in breakdown.items()}
        if other_segments is not None:
            # (the forms `value` is NOT: "default forms" when the selected ones won, the selected ones when the defaults stayed)
            out["segments_us_other_forms"] = {"forms": loser["forms"] or "", "segments_us": other_segments}
        if selection is not None:
            # every candidate's us/step and whether its results were bit-identical to the default forms' on the verified frames
            if isinstance(selection.get("table"), list):      # (the forms in effect of every candidate stay in the child's log: the line keeps the verdicts)
                selection = dict(selection, table=[{k: v for k, v in m.items() if k != "effective"} for m in selection["table"] if isinstance(m, dict)])
            out["form_selection"] = selection
        if not isinstance(late_errors, dict) and late_errors != 0:
            out["device_errors_after_timed_region"] = late_errors
        Mh = max(1, counts.get("n_hi", M) or M)
        if filt_ok:
            if hasattr(trk, "_filter_raw") and os.environ.get("BTS_BENCH_IGNORE_ERRORS"):
                filt = dict(filt, raw=[int(v) for v in trk._filter_raw])
            out["prefilter"] = dict(filt, full_k_stages_equivalent=int(-(-K // filt["tile"][0]) * -(-Mh // filt["tile"][1]) * (D // 128)),
                                    candidates_per_high_detection=filt["candidates"] / Mh,
                                    what="last step: K stages of 128 elements run over all tiles (`tile` rows x columns), tiles that stopped at a prefix "
                                         "checkpoint, pairs handed to the exact float32 refinement")
        else:
            out["prefilter"] = filt

        def lap_info():
            d = {"stage%d" % s: dict(zip(("edges", "components", "overflow", "hard_edges", "form"),
                                         (lambda h: (int(h[0]), int(h[1]), int(h[2]), int(h[6]), int(h[8])))(trk.debug_lap(s)))) for s in (1, 2, 3)}
            d["stage1_us"] = round(breakdown.get("k_lap", 0.0), 2)      # the stage-1 launch alone (hipEvent pair around it)
            d["stage23_group_us"] = round(breakdown.get("stage2", 0.0), 2)   # k_lap(stages 2 + 3) + tk_apply23
            return d
        out["lap"] = guarded("lap", lap_info)
        if export is not None:
            out["export_inclusive"] = export
        if facade is not None:
            out["facade"] = facade
        if id_multi is not None:
            def gather_ids():
                bad = [r for r in id_multi if not isinstance(r, dict) or "error" in r]
                if bad:
                    return {"error": "rank checker failed: %s" % str(bad[0])[:200], "streams": world}
                keys = ("frames", "frames_identical", "assignment_pairs", "assignment_pairs_equal", "track_ids", "track_ids_equal", "device_errors")
                d = {k: sum(r[k] for r in id_multi) for k in keys}
                d.update({"kalman_mean_max_abs_err": max(r["kalman_mean_max_abs_err"] for r in id_multi), "kalman_tolerance": 1e-4,
                          "streams": world, "frames_identical_per_stream": [r["frames_identical"] for r in id_multi],
                          "stream_seeds": [r.get("stream_seed") for r in id_multi],
                          "against": "CPU oracle, the first frames of every rank's own stream"})
                return d
            out["id_match"] = guarded("id_match", gather_ids)
        if not a.no_cpu and world == 1:
            if not big and not a.no_standalone:
                # the dense kernels of the path, standalone, against their rooflines (a few seconds; at small sizes the
                # step above is launch-bound, these show what the kernels do when they have work)
                out["standalone_rooflines"] = guarded("standalone_rooflines", lambda: importlib.import_module("bot-sort-onnx-tensorrt_amd.microbench").run(full=False))
            if a.several_streams and S == 1 and not a.graph:
                # what ONE GPU serves when it is given several independent streams of this workload (own handle, HIP stream
                # and host thread each: multistream.run_concurrent); not the headline configuration (1 stream per GPU)
                out["several_streams_one_gpu"] = several_streams(a, (4, 8))
            checker = guarded("id_match", IdMatch, pkg, K, M, D, a.motion_gate, forms=forms) if (S == 1 and not a.dense) else None
            checker_err = checker if isinstance(checker, dict) else None
            if checker_err:
                checker = None
            if not big:
                out["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), a.cpu_seconds, checker, motion_gate=a.motion_gate)
                if isinstance(out["cpu_baseline"], dict) and "error" in out["cpu_baseline"] and checker is not None:
                    # the checker raised inside the timed oracle loop: the baseline itself is re-taken without it
                    out["id_match"] = {"error": out["cpu_baseline"]["error"]}
                    checker = None
                    out["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), a.cpu_seconds, None, motion_gate=a.motion_gate)
                if isinstance(out.get("cpu_baseline"), dict):
                    rv = guarded("cpu_baseline.vectorised", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), max(3.0, a.cpu_seconds / 3), None, faithful=False, motion_gate=a.motion_gate)
                    out["cpu_baseline"]["vectorised"] = {k: rv.get(k) for k in ("value", "unit", "cores", "sample", "error") if k in rv} if isinstance(rv, dict) else None
            else:
                # big tables: the structure-faithful oracle takes seconds to minutes per frame -- it is timed alone on what
                # fits the budget, and the ID match runs on 4 steady-state frames of the vectorised form of the same oracle
                out["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), a.cpu_seconds, None, motion_gate=a.motion_gate)
                r = None
                if checker is not None:
                    r = guarded("id_match", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), 120.0, checker, faithful=False, max_frames=4, motion_gate=a.motion_gate)
                    if isinstance(r, dict) and "error" in r:
                        out["id_match"] = {"error": r["error"]}
                        checker = None
                        r = None
                if r is None:
                    r = guarded("cpu_baseline.vectorised", cpu_baseline, pkg, a.workload, ms_seed(pkg, 0), max(5.0, a.cpu_seconds / 2), None, faithful=False, max_frames=4, motion_gate=a.motion_gate)
                if isinstance(out.get("cpu_baseline"), dict):
                    # BASELINE.md section 3's second CPU figure: the vectorised form of the oracle on the same stream (its timed
                    # brackets exclude the checker's device work)
                    out["cpu_baseline"]["vectorised"] = {k: r.get(k) for k in ("value", "unit", "cores", "sample", "error") if k in r} if isinstance(r, dict) else None
            if checker is not None:
                out["id_match"] = guarded("id_match", checker.result)
            elif checker_err:
                out["id_match"] = checker_err
        if world == 1 and S == 1 and not a.no_cpu and not os.environ.get("BTS_LIB_PATH") and not a.graph and not a.dense:
            tl = guarded("trace_legs", trace_legs, a.workload)
            if tl:
                out["trace_variants"] = tl
        guard.emit(out)
    elif guard.timer is not None:
        guard.timer.cancel()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
