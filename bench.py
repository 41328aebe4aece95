#!/usr/bin/env python
"""Tracker-only benchmark: BoT-SORT update steps / second on pre-extracted detections
(BASELINE.json metric), one independent video stream per GPU (no data-path collective).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|target|c5]

A "step" is one whole BoTSORT.update (tracking half) = one `bts_tracker_step` on one frame of the
seeded synthetic stream.  Frames are generated on the host and uploaded to HBM
before the timed region.  Rank 0 prints ONE JSON line (contract in the task statement) with two
extra objects:
  roofline      the dominant kernel group of the step, timed live with hipEvents on the step's
                stream inside the timed region (C ABI bts_tracker_profile_*), against its bound
  cpu_baseline  the CPU oracle (structure-faithful port of the reference: per-pair Python IoU,
                scalar Kalman updates, dense extended-matrix JV) on the same stream, 1 core,
                bounded to ~15 s.
  id_match      the metric's "assignment ID match vs ref": a second tracker handle stepped on the
                frames of the cpu_baseline sample, compared with the oracle frame by frame
                (assignment pairs of the three stages, ordered track ids / states, Kalman means).
For N > 1 launch with torch.distributed.run; ranks only meet at barriers (gloo control plane).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F8_MFMA_PEAK_TF = 5000.0       # dense fp8 MFMA peak (not the 2:1-sparse headline): the pre-filter multiplies fp8 shadows
F32_MFMA_PEAK_TF = 157.3       # dense f32-input MFMA peak (= f32 vector peak)
SEGMENTS = ("prep", "cos1", "cost1", "lap1", "apply1", "stage2", "merge", "dedupe")   # launch groups of a step
KERNELS = {   # single kernels that can dominate a step -> name in the rocprofv3 kernel trace
    "k_filter": "k_filter_dma<false> | k_filter_dma_p<false>", "k_lap": "k_lap<4|8>", "k_edges1": "tk_edges<1,1,*> | tk_edges_tall<*> | tk_edges_tall_gate<*> | tk_edges_tall_gate_c<*>", "k_mid2": "tk_mid2<1>",
    "k_apply23": "tk_apply23", "prep": "tk_front", "merge": "tk_merge",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--segment", default="auto", help="kernel timed for the roofline object (auto = the dominant one)")
    ap.add_argument("--event-stride", type=int, default=8, help="hipEvent pairs around the roofline kernel on every n-th timed step "
                    "(a pair costs the stream ~4 us: on every launch it inflates a 90 us step by several percent)")
    ap.add_argument("--streams-per-gpu", type=int, default=1, help="independent video streams per GPU, each on its own HIP stream and "
                    "host thread (the headline metric is 1; more shows what a GPU serves when steps are launch-bound)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dense", action="store_true", help="dense float32 appearance path instead of the f16 pre-filter")
    ap.add_argument("--graph", action="store_true", help="replay each step as one hipGraph (bts_tracker_config.use_graph)")
    ap.add_argument("--breakdown", action="store_true", help="also report the per-segment timing table")
    ap.add_argument("--motion-gate", action="store_true", help="Mahalanobis gate on the stage-1 pairs (bts_tracker_config.motion_gate; "
                    "off = the reference's cascade).  BASELINE config 3 is stated with the gate on: --workload c3 implies it")
    ap.add_argument("--no-motion-gate", action="store_true", help="c3 without the gate")
    return ap.parse_args()


def kernel_model(kern, K, M, D, dense):
    """Algorithmic work of ONE launch of a kernel in a steady-state frame (pool N = K rows, M detections,
    7/8 of them high-score, ~M matches, ~1.2 feasible edges per detection): (bound, amount, unit, peak).
    Formulas: DESIGN.md section 3."""
    N, Mh, Km = K, M - M // 8, M - M // 8
    E = int(1.2 * Mh)
    if kern == "k_filter":     # appearance contraction 2*N*Mh*D flop on the fp8 matrix cores
        return "mfma", 2.0 * N * Mh * D, "flop", F8_MFMA_PEAK_TF
    table = {
        # stage-1 matches: Kalman rows read + written, measurement read; det row, smooth r+w, curr w, f16 shadow w
        "k_mid2": Km * (2 * 72 + 4) * 8 + Km * D * (4 + 4 + 4 + 4 + 2),
        "k_apply23": (M // 8) * ((2 * 72 + 4) * 8 + D * 18),    # the low-score matches of stage 2 (stage 3 is empty in steady state)
        "k_edges1": (N + Mh) * 32 + E * 16 + Mh * 2 * D * 4,    # boxes in, sparse edges out, two feature rows per refined candidate
        "k_lap": E * 16 + 8 * Mh,                               # edge list in, assignment vectors out
        "prep": M * D * 6 + M * 48,                             # detection rows read (f32) + f16 shadows written, boxes
        "merge": (N + M) * 16,                                  # list entries and per-slot flags
    }
    return "hbm", float(table[kern]), "byte", HBM_PEAK_GBS


class IdMatch:
    """Checker beside the CPU-baseline leg (the metric's "assignment ID match vs ref"): a second tracker handle is
    stepped on the very frames the oracle is timed on (outside the oracle's timed bracket) and, per frame, the three
    stages' assignment indices, the ordered track ids / states of the tracked and lost lists and the Kalman means are
    compared with the oracle's."""
    INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")

    def __init__(self, pkg, K, M, D, motion_gate=False):
        import torch
        self.torch = torch
        self.trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(),
                                     motion_gate=1 if motion_gate else 0)
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
                "against": "CPU oracle (pinned to the reference's golden vectors), every frame of the cpu_baseline sample incl. "
                           "the two pool-building frames; tracked + lost lists in order, stages 1-3 (row, column) pairs"}


def cpu_baseline(pkg, K, M, D, seed, budget_s, checker=None, faithful=True, max_frames=200, motion_gate=False):
    """The oracle timed on this box's host cores: 1 process, 1 core (the reference is
    single-threaded Python).  Only this leg may touch oracle/."""
    from oracle import bts_oracle as bo
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    st = pkg.synth.SyntheticStream(K, M, D, seed=seed)
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
    return {"value": n / t_used, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": "%d steady-state frames (frames 2..%d) of the same seeded stream, %.1f s, structure-faithful oracle "
                      "(per-pair Python IoU, scalar Kalman update, dense JV on the (N+M)^2 matrix); host has %d cores"
                      % (n, n + 1, t_used, os.cpu_count() or 0)}


def pmc_traffic(workload, kernel_name):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE,
    separate runs: scripts/gpu_final.sh -> scripts/pmc_to_json.py).  None when no pass covers this workload."""
    try:
        tab = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))["workloads"][workload]
    except Exception:
        return None
    bases = [alt.strip().split("<")[0] for alt in kernel_name.split("|")]      # (a kernel may have several forms)
    best = None
    for k, v in tab.items():
        if k.split("<")[0] in bases and v.get("traffic_bytes") is not None:
            best = max(best or 0.0, v["traffic_bytes"])       # the form that ran (the others are absent or tiny)
    return best


def several_streams(a, counts):
    """Aggregate steps/s of n concurrent streams on this GPU, n in `counts`: `bench.py --streams-per-gpu n`, each in a
    fresh process (the runtime binds HIP streams to its hardware queues at first use; a process that has already used
    other streams gets less regular bindings)."""
    import subprocess
    out = []
    for n in counts:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", a.workload, "--steps", str(min(a.steps, 200)),
               "--warmup", str(a.warmup), "--no-cpu", "--streams-per-gpu", str(n)] + (["--dense"] if a.dense else []) + (["--motion-gate"] if a.motion_gate else ["--no-motion-gate"])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out.append({"streams": n, "value": d["value"], "unit": d["unit"], "steps_per_stream": d["steps"]})
        except Exception as e:          # a reported extra, never fatal for the headline line
            out.append({"streams": n, "value": None, "error": str(e)[:200]})
    return out


def ms_seed(pkg, rank):
    return pkg.multistream.stream_seed(1234, rank)      # one independent video stream per GPU


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required (no CPU fallback)"
    torch.cuda.set_device(local % torch.cuda.device_count())
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    K, M, D = pkg.synth.CONFIGS[a.workload]
    a.motion_gate = (a.motion_gate or a.workload == "c3") and not a.no_motion_gate
    a.warmup = max(a.warmup, 3)          # (frames 0-1 build the track pool, frame 2 is the first steady-state one: never timed)

    S = max(1, a.streams_per_gpu)
    st = pkg.synth.SyntheticStream(K, M, D, seed=ms_seed(pkg, rank * S))
    PROBE = 6

    def next_frame():
        fr = st.next()
        return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())

    stream = torch.cuda.Stream()          # graph replay needs a real (non-legacy) stream
    torch.cuda.set_stream(stream)
    frames = [next_frame() for _ in range(a.warmup)]
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(),
                            appearance_mode=1 if a.dense else 0, use_graph=1 if a.graph else 0, motion_gate=1 if a.motion_gate else 0)

    ms = importlib.import_module("bot-sort-onnx-tensorrt_amd.multistream")
    ddist = dist if world > 1 else None

    for k in range(a.warmup):
        trk.step(*frames[k])
    torch.cuda.synchronize()
    # untimed probe (more warm-up): which kernel group dominates a step?
    breakdown = {}
    for seg in SEGMENTS + tuple(k for k in KERNELS if k not in SEGMENTS) + ("step",):
        trk.profile_enable(seg, PROBE)
        for _ in range(PROBE):
            trk.step(*next_frame())
        ms_, n_ = trk.profile_read()
        breakdown[seg] = 1e3 * ms_ / max(n_, 1)
    segment = a.segment if a.segment != "auto" else max(KERNELS, key=lambda k: breakdown[k])
    trk.profile_enable(segment, 0 if a.graph else a.steps, max(1, min(a.event_stride, a.steps // 16)))     # (>= 16 samples)     # graph replay cannot carry per-kernel events
    timed = [next_frame() for _ in range(a.steps)]
    extra = []                       # further streams of this GPU: (tracker, HIP stream, frames), warmed up the same way
    for k in range(1, S):
        st_k = pkg.synth.SyntheticStream(K, M, D, seed=ms_seed(pkg, rank * S + k))
        hs = None
        with torch.cuda.stream(torch.cuda.default_stream()):      # (set-up on the default stream: see multistream.run_concurrent)
            tk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D, device=torch.cuda.current_device(),
                                   appearance_mode=1 if a.dense else 0, use_graph=1 if a.graph else 0, motion_gate=1 if a.motion_gate else 0)
            for _ in range(a.warmup + 6):
                fr = st_k.next()
                tk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
            fl = []
            for _ in range(a.steps):
                fr = st_k.next()
                fl.append((torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda()))
        extra.append((tk, hs, fl))
    if a.graph:                                              # capture happens on the first replayed step: keep it untimed
        trk.step(*next_frame())
        torch.cuda.synchronize()

    ms.barrier(ddist)
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
    ms.barrier(ddist)
    if a.graph:      # the dominant kernel's duration: same number of further steps, eager, with events
        trk.profile_enable(segment, a.steps)
        for _ in range(a.steps):
            trk.step(*next_frame())
        torch.cuda.synchronize()
    seg_ms, seg_n = trk.profile_read()
    counts = trk.counts()
    assert counts["errors"] == 0, counts
    for e in extra:
        assert e[0].counts()["errors"] == 0
    elapsed = ms.max_over_ranks(t1 - t0, ddist)
    if a.breakdown and rank == 0:
        sys.stderr.write("segment us/step: " + json.dumps({k: round(v, 1) for k, v in breakdown.items()}) + "\n")

    id_multi = None
    if world > 1 and not a.no_cpu and S == 1 and not a.dense:
        # ID match of EVERY rank's stream (N > 1: no CPU baseline, that is the N = 1 line's): 24 steady-state frames of this
        # rank's stream through a second handle and the oracle (vectorised IoU form: the checker, not a timing), gathered
        chk = IdMatch(pkg, K, M, D, a.motion_gate)
        cpu_baseline(pkg, K, M, D, ms_seed(pkg, rank), 60.0, chk, faithful=False, max_frames=24, motion_gate=a.motion_gate)
        id_multi = [None] * world
        dist.all_gather_object(id_multi, chk.result())

    if rank == 0:
        bound, amount, unit, peak = kernel_model(segment, K, M, D, a.dense)
        seg_s = (seg_ms / 1e3) / max(seg_n, 1)
        if bound == "mfma":
            achieved, runit = amount / seg_s / 1e12, "TFLOP/s"
        else:
            achieved, runit = amount / seg_s / 1e9, "GB/s"
        out = {
            "metric": "tracker update-steps/sec (N_tracks x M_dets)",
            "value": ms.aggregate_throughput(a.steps * S, elapsed, world),
            "unit": "steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,      # (one step of every stream)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded stream generator, SURVEY.md 8d; frames resident in HBM before the timed region)",
            "config": {"workload": "%s: %d tracks x %d detections/frame x %d-d ReID, IoU+cosine fused cost, "
                                   "3-stage association%s, one stream per GPU" % (a.workload, K, M, D, ", Mahalanobis gate on" if a.motion_gate else ""),
                       "tracks": K, "dets": M, "feat_dim": D, "streams": world * S, "parallelism": "%d stream%s/GPU, no collective" % (S, "" if S == 1 else "s (own HIP stream + host thread each)"),
                       "launch": "hipGraph replay" if a.graph else "eager", "motion_gate": bool(a.motion_gate)},
            "roofline": {"kernel": KERNELS[segment], "bound": bound, "achieved": achieved, "peak": peak, "unit": runit,
                         "frac": achieved / peak, "traffic": pmc_traffic(a.workload, KERNELS[segment]),
                         "traffic_unit": "bytes per launch (HBM, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, profiles/pmc_traffic.json)",
                         "avg_us": 1e6 * seg_s, "launches_timed": seg_n, "event_stride": max(1, min(a.event_stride, a.steps // 16)),
                         "algorithmic_per_launch": amount, "algorithmic_unit": unit},
        }
        out["segments_us"] = {k: round(v, 2) for k, v in breakdown.items()}
        if id_multi is not None:
            keys = ("frames", "frames_identical", "assignment_pairs", "assignment_pairs_equal", "track_ids", "track_ids_equal", "device_errors")
            out["id_match"] = {k: sum(r[k] for r in id_multi) for k in keys}
            out["id_match"].update({"kalman_mean_max_abs_err": max(r["kalman_mean_max_abs_err"] for r in id_multi), "kalman_tolerance": 1e-4,
                                    "streams": world, "frames_identical_per_stream": [r["frames_identical"] for r in id_multi],
                                    "against": "CPU oracle, the first 26 frames of every rank's own stream"})
        if not a.no_cpu and world == 1:
            # the dense kernels of the path, standalone, against their rooflines (a few seconds; the step above is
            # launch-bound at this size, these show what the kernels do when they have work)
            out["standalone_rooflines"] = importlib.import_module("bot-sort-onnx-tensorrt_amd.microbench").run(full=False)
            if world == 1 and S == 1 and not a.graph:
                # what ONE GPU serves when it is given several independent streams of this workload (own handle, HIP stream
                # and host thread each: multistream.run_concurrent); not the headline configuration (1 stream per GPU)
                out["several_streams_one_gpu"] = several_streams(a, (4, 8))
            checker = IdMatch(pkg, K, M, D, a.motion_gate) if (S == 1 and not a.dense) else None
            if K <= 1024:
                out["cpu_baseline"] = cpu_baseline(pkg, K, M, D, ms_seed(pkg, 0), a.cpu_seconds, checker, motion_gate=a.motion_gate)
            else:
                # big tables: the structure-faithful oracle takes seconds to minutes per frame -- it is timed alone on what
                # fits the budget, and the ID match runs on 6 steady-state frames of the vectorised form of the same oracle
                out["cpu_baseline"] = cpu_baseline(pkg, K, M, D, ms_seed(pkg, 0), a.cpu_seconds, None, motion_gate=a.motion_gate)
                if checker is not None:
                    cpu_baseline(pkg, K, M, D, ms_seed(pkg, 0), 120.0, checker, faithful=False, max_frames=6, motion_gate=a.motion_gate)
            if checker is not None:
                out["id_match"] = checker.result()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
