"""(File name: sorts after the suites that have been green on hardware, like the other test_zz_gpu_* files.)

Round-4 GPU parity tests -- written, like round 3's, while GPU access was closed, and run on the CPU emulation of the kernel
sources (tests/hip_emu, which since this round executes k_filter_i8 / k_filter_splitk / k_cos_mfma themselves at
instruction level):
  * BASELINE config 5's arithmetic mode (kalman_dtype BTS_F32, cost_dtype BTS_F16): ID-match RATE against the float64 oracle,
    Kalman state within north_star's 1e-4;
  * the two forms of the assignment kernel (BTS_LAP_FORM=row | col) give the same assignment;
  * the persistent one-pass pre-filter with its prefix exits, at sizes an emulation can run (BTS_FILTER_ONEPASS), for both
    checkpoint sets (BTS_I8_FIRST16 = 5: round 2's; default: derived from the gate);
  * bts_tracker_inputs_resident + BTS_CVT_OVERLAP (side-stream conversion with and without the promise)."""
import ctypes
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")
BTS_F64, BTS_F32, BTS_F16 = 0, 1, 2


@pytest.fixture(scope="module")
def bts():
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    pkg._lib.load(require_gpu=True)
    return pkg


def _dev(fr):
    import torch
    return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())


def _id_match_rate(bts, st, K, M, D, frames, **opts):
    """Frames whose three assignments and ordered track tables equal the float64 oracle's; worst Kalman-mean error over the
    frames that agree, relative to the track's POSITION scale max(1, |x|, |y|, |w|, |h|) -- the magnitude that sets the float32
    quantum of the whole state (the gain mixes position residuals into the velocities: a velocity of ~1 px/frame carries the
    ~1e-4 px absolute rounding of a coordinate of ~2000 px)."""
    from oracle import bts_oracle as bo
    ora = bo.TrackerOracle(faithful_iou=False)
    trk = bts.DeviceTracker(max_tracks=K + max(256, K // 2), max_dets=max(K, M), feat_dim=D, **opts)
    same, worst, pairs, pairs_eq = 0, 0.0, 0, 0
    for f in range(frames):
        fr = st.next()
        trk.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        ok = True
        for k in (1, 2, 3):
            got, want = np.asarray(trk.stage_result(k)["m"]).reshape(-1, 2), np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)
            eq = got.shape == want.shape and bool((got == want).all())
            pairs += len(want)
            pairs_eq += len(want) if eq else len(set(map(tuple, got)) & set(map(tuple, want)))
            ok &= eq
        for which, lst in ((0, ora.tracked), (1, ora.lost)):
            g, w = trk.export(which), bo.TrackerOracle.table(lst)
            tab_ok = all(np.array_equal(g[key], w[key]) for key in INT_KEYS)
            ok &= tab_ok
            if tab_ok and len(w["track_id"]):
                scale = np.maximum(1.0, np.abs(w["mean"][:, :4]).max(axis=1, keepdims=True))
                worst = max(worst, float((np.abs(g["mean"] - w["mean"]) / scale).max()))
        same += int(ok)
    assert trk.counts()["errors"] == 0
    trk.close()
    return same, worst, pairs, pairs_eq


@pytest.mark.parametrize("K,M,D,frames", [(300, 150, 384, 24), (1638, 410, 384, 6)])
def test_config5_arithmetic_mode_id_match_rate(bts, K, M, D, frames):
    """BASELINE configs[4] "fp32 Kalman / fp16 cost matrix" as an opt-in mode: the state is stored in
    float32, every association cost is rounded to binary16.  Not bit-parity -- the bar is an ID-match RATE against the float64
    oracle on a config-5-shaped stream (384-d features, 4 tracks per detection) and north_star's 1e-4 on the Kalman state.
    (scripts/c5_precision_study.py predicted 130 / 130 identical frames on the CPU.)"""
    synth = bts.synth
    same, worst, pairs, pairs_eq = _id_match_rate(bts, synth.SyntheticStream(K, M, D, seed=1234, churn=0.05), K, M, D, frames,
                                                  kalman_dtype=BTS_F32, cost_dtype=BTS_F16)
    assert pairs > 2 * M
    assert pairs_eq >= 0.995 * pairs, (pairs_eq, pairs)
    assert same >= frames - max(1, frames // 20), (same, frames)
    assert worst <= 1e-6, worst                 # (north_star's bar on the state is 1e-4; float32 storage leaves ~1e-7 of the state's scale)
    # the default mode on the same stream stays exact (the mode is a switch, not a change of the default arithmetic)
    same64, worst64, _, _ = _id_match_rate(bts, synth.SyntheticStream(K, M, D, seed=1234, churn=0.05), K, M, D, min(frames, 6))
    assert same64 == min(frames, 6) and worst64 <= 1e-12


def test_config5_arithmetic_mode_full_size(bts, c5_oracle_frames):
    """The mode at config 5's own size, 16384 tracks x 4096 detections x 384-d: three frames against the vectorised oracle (its run
    is shared with tests/test_gpu_round2.py::test_config5_stress_size_vs_oracle: conftest.c5_oracle_frames)."""
    import torch
    K, M, D = 16384, 4096, 384
    trk = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D, kalman_dtype=BTS_F32, cost_dtype=BTS_F16)
    same, worst, pairs, pairs_eq = 0, 0.0, 0, 0
    for fr in c5_oracle_frames:
        trk.step(torch.from_numpy(fr["boxes"]).cuda(), torch.from_numpy(fr["scores"]).cuda(), torch.from_numpy(fr["feats"].copy()).cuda())
        ok = True
        for k in (1, 2, 3):
            got, want = np.asarray(trk.stage_result(k)["m"]).reshape(-1, 2), fr["m"][k]
            eq = got.shape == want.shape and bool((got == want).all())
            pairs += len(want)
            pairs_eq += len(want) if eq else len(set(map(tuple, got)) & set(map(tuple, want)))
            ok &= eq
        for which in (0, 1):
            g, w = trk.export(which), fr["tables"][which]
            tab_ok = all(np.array_equal(g[key], w[key]) for key in INT_KEYS)
            ok &= tab_ok
            if tab_ok and len(w["track_id"]):
                scale = np.maximum(1.0, np.abs(w["mean"][:, :4]).max(axis=1, keepdims=True))
                worst = max(worst, float((np.abs(g["mean"] - w["mean"]) / scale).max()))
        same += int(ok)
    assert trk.counts()["errors"] == 0
    trk.close()
    assert pairs > 3500 and pairs_eq >= 0.995 * pairs and same >= 2 and worst <= 1e-6, (same, worst, pairs, pairs_eq)


def test_config5_mode_halves_the_kalman_bytes_and_is_validated(bts):
    lib, L = bts._lib.load(), bts._lib
    cfg = L.TrackerConfig()
    assert lib.bts_tracker_default_config(ctypes.byref(cfg)) == 0
    assert cfg.kalman_dtype == BTS_F64 and cfg.cost_dtype == BTS_F64          # the default is the reference's arithmetic
    cfg.max_tracks, cfg.max_dets, cfg.feat_dim = 16384, 4096, 384
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert lib.bts_tracker_memory_bytes(ctypes.byref(cfg), ctypes.byref(a)) == 0
    cfg.kalman_dtype = BTS_F32
    assert lib.bts_tracker_memory_bytes(ctypes.byref(cfg), ctypes.byref(b)) == 0
    assert 0.98 * 16384 * 72 * 4 <= a.value - b.value <= 16384 * 72 * 4 + 1024          # 72 state words per track: 8 -> 4 bytes
    with pytest.raises(bts.BtsError):
        bts.DeviceTracker(max_tracks=64, max_dets=32, feat_dim=128, kalman_dtype=7)
    with pytest.raises(bts.BtsError):
        bts.DeviceTracker(max_tracks=64, max_dets=32, feat_dim=128, cost_dtype=BTS_F16, tie_mode=1)
    # the per-slot reads widen the float32 rows (the call's contract is float64)
    import torch
    trk = bts.DeviceTracker(max_tracks=64, max_dets=32, feat_dim=128, kalman_dtype=BTS_F32)
    fr = bts.synth.SyntheticStream(16, 16, 128, seed=3).next()
    trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())
    t = trk.export(0)
    assert len(t["track_id"]) > 0
    slot = int(t["slot"][0])
    mean, cov = trk.read_slot(slot, 0), trk.read_slot(slot, 1)
    assert np.array_equal(mean, t["mean"][0]) and np.array_equal(mean, mean.astype(np.float32).astype(np.float64))
    assert cov.shape == (8, 8) and cov[0, 0] > 0 and np.array_equal(cov, cov.astype(np.float32).astype(np.float64))
    trk.close()


@pytest.mark.parametrize("form", ["col", "row", "col+self", "row+self", "col+self3"])
@pytest.mark.parametrize("name", ["crowd@c2", "plain"])
def test_solver_forms_agree(bts, monkeypatch, name, form):
    """k_lap<EC, CROWD>: the column forms (default) and the round-2 row forms (BTS_LAP_FORM=row when the tracker is created)
    both return the oracle's assignment -- on a plain scene and on a dense crowd (components of 100+ rows).  "+self": the SELF
    form of the launch (BTS_LAP_SELF=1: every workgroup prepares the problem itself, components ranked by their smallest row id
    and dealt round-robin; "self3": with three helper workgroups only, so that every wavefront takes several components)."""
    if "self" in form:
        monkeypatch.setenv("BTS_LAP_SELF", "1")
        if form.endswith("3"):
            monkeypatch.setenv("BTS_LAP_HELPERS", "3")
        form = form.split("+")[0]
    synth = bts.synth
    from oracle import bts_oracle as bo
    if name == "plain":
        K, M, D, st = 512, 256, 256, synth.SyntheticStream(512, 256, 256, seed=9, churn=0.05)
    else:
        K, M, D = synth.workload_dims(name)
        st = synth.make_workload(name, seed=77)
    ora = bo.TrackerOracle(faithful_iou=False)
    monkeypatch.setenv("BTS_LAP_FORM", form)
    trk = bts.DeviceTracker(max_tracks=K + 256, max_dets=max(K, M), feat_dim=D)
    want_self = "1" if "BTS_LAP_SELF" in __import__("os").environ else "0"
    assert trk.forms()["lap_self"] == want_self and trk.forms()["lap_form"] == form
    for v in ("BTS_LAP_FORM", "BTS_LAP_SELF", "BTS_LAP_HELPERS"):
        monkeypatch.delenv(v, raising=False)
    hard, modes = 0, set()
    for f in range(5):
        fr = st.next()
        trk.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            assert np.array_equal(trk.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        hard += int(trk.debug_lap(1)[1])
        if f >= 2:                        # which preparation ran (hdr word 8): 1 = LDS form, classic hand-over; 3 = SELF form
            mode = int(trk.debug_lap(1)[8])   # (2 = the global-memory preparation took over -- the row forms on the crowd: every
            assert mode == 2 or mode == (3 if want_self == "1" else 1), (f, trk.debug_lap(1)[:10])      # workgroup of a SELF launch then falls back to the classic hand-over)
            modes.add(mode)
    assert trk.counts()["errors"] == 0
    assert hard > 0                       # (non-trivial components were solved, not only isolated pairs)
    assert (2 in modes) == (name.startswith("crowd") and form == "row"), modes
    trk.close()


@pytest.mark.parametrize("first16", [None, "5", "8", "guess", "guess8", "planes", "planes+guess"])
def test_one_pass_prefilter_prefix_exits_vs_dense(bts, monkeypatch, first16):
    """The persistent one-pass form of the pre-filter (k_filter_i8: tile walk, LDS-DMA double buffering, prefix checkpoints,
    exit prediction, the next tile's first stage issued under the test) at a size small enough for a CPU emulation:
    BTS_FILTER_ONEPASS sends a 384 x 256 x 2048 problem (3 x 2 tiles, 16 K stages, checkpoints in use) through it instead of
    through split K.  Cosines sit within +-0.01 of the gate, boxes teleport (appearance alone re-associates): FILTER must equal
    DENSE frame by frame, for the round-2 checkpoint set (5), the gate-derived default and a late first checkpoint (8)."""
    import torch
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    if first16 is not None and first16.startswith("planes"):         # BTS_SHADOW_PLANES: the int8 shadows in K-stage planes (16 planes here)
        monkeypatch.setenv("BTS_SHADOW_PLANES", "1")
        first16 = first16[7:] or None
    planes = "BTS_SHADOW_PLANES" in __import__("os").environ
    guess = first16 is not None and first16.startswith("guess")      # BTS_FILTER_GUESS: a workgroup's first tile is predicted to stop
    if guess:                                                        # at the first checkpoint as well (changes prefetches only); with
        monkeypatch.setenv("BTS_FILTER_GUESS", "1")                  # a late first checkpoint the prediction is sometimes wrong
        monkeypatch.setenv("BTS_FILTER_PW", "2")                     # (two persistent workgroups: three tiles each, so the
        first16 = first16[5:] or None                                #  next-tile prefetch of a first tile is exercised)
    if first16 is not None:
        monkeypatch.setenv("BTS_I8_FIRST16", first16)
    rng = np.random.default_rng(41)
    K, D = 256, 2048
    proto = rng.standard_normal((K, D))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    a = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D)
    assert a.forms()["filter_guess"] == ("1" if guess else "0") and a.forms()["shadow_planes"] == ("1" if planes else "0")
    b = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D, appearance_mode=1)
    early, matched1 = 0, 0
    for f in range(4):
        cx = rng.permutation(K) * 7 + 10
        cy = rng.integers(100, 8000, K)
        boxes = np.stack([cx, cy, cx + 5, cy + 9], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, K) if f == 0 else rng.uniform(0.5, 0.85, K)).astype(np.float32)
        c = 0.75 + rng.uniform(3e-5, 0.01, K) * rng.choice([-1.0, 1.0], K)
        if f == 0:
            c[:] = 1.0
        noise = rng.standard_normal((K, D))
        noise -= (noise * proto).sum(1, keepdims=True) * proto
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        feats = (c[:, None] * proto + np.sqrt(1 - c[:, None] ** 2) * noise).astype(np.float32)
        proto = feats.astype(np.float64) / np.linalg.norm(feats.astype(np.float64), axis=1, keepdims=True)
        perm = rng.permutation(K)
        boxes, scores, feats, proto = boxes[perm], scores[perm], np.ascontiguousarray(feats[perm]), proto[perm]
        fr = (torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda())
        a.step(*fr, torch.from_numpy(feats.copy()).cuda())
        b.step(*fr, torch.from_numpy(feats.copy()).cuda())
        for k in (1, 2, 3):
            ra, rb = a.stage_result(k), b.stage_result(k)
            assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["y"], rb["y"]), (f, k)
        for which in (0, 1):
            ta, tb = a.export(which), b.export(which)
            for key in INT_KEYS:
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
        if f == 1:
            matched1 = len(a.stage_result(1)["m"])
        if f >= 1:
            st = a.debug_filter()
            assert st["tiles"] >= 2 and st["k_stages"] >= st["tiles"] * 3
            early += st["prefix_exits"]
            assert st["candidates"] >= len(a.stage_result(1)["m"])          # every appearance match was a candidate
    assert a.counts() == b.counts() and a.counts()["errors"] == 0
    assert matched1 > K // 4           # about half of frame 1's cosines sit inside the gate
    assert early > 0                   # tiles did stop at a prefix (random 2048-d rows: nothing but the true matches survives)
    a.close(); b.close()


@pytest.mark.parametrize("resident", [False, True])
def test_cvt_overlap_with_and_without_the_residency_promise(bts, monkeypatch, resident):
    """BTS_CVT_OVERLAP (side-stream conversion of the detection rows): ordered behind the caller's stream by an event unless
    bts_tracker_inputs_resident promised complete inputs.  Results are the default path's either way."""
    synth = bts.synth
    K, M, D = 600, 300, 256
    ref = bts.DeviceTracker(max_tracks=K + 128, max_dets=max(K, M), feat_dim=D)
    monkeypatch.setenv("BTS_CVT_OVERLAP", "1")
    exp = bts.DeviceTracker(max_tracks=K + 128, max_dets=max(K, M), feat_dim=D)
    monkeypatch.delenv("BTS_CVT_OVERLAP")
    exp.inputs_resident(resident)
    st = synth.SyntheticStream(K, M, D, seed=12, churn=0.05)
    for f in range(6):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        for k in (1, 2, 3):
            assert np.array_equal(ref.stage_result(k)["m"], exp.stage_result(k)["m"]), (f, k)
        for which in (0, 1):
            ta, tb = ref.export(which), exp.export(which)
            for key in INT_KEYS:
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"])
    assert ref.counts() == exp.counts() and ref.counts()["errors"] == 0
    ref.close(); exp.close()


def _sorted_vs_default(bts, monkeypatch, frames_fn, K_tab, M_cap, D, nframes, oracle=True, onepass=True):
    """the same frames through a default tracker and one created with BTS_EDGES_SORTED=1 (+ BTS_FILTER_ONEPASS so that small
    problems take the one-pass pre-filter the sorted form builds on): stage results, track tables and Kalman means must be
    IDENTICAL frame by frame (the edge SET is the same, only its order differs), and equal to the oracle's."""
    from oracle import bts_oracle as bo
    if onepass:
        monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    ref = bts.DeviceTracker(max_tracks=K_tab, max_dets=M_cap, feat_dim=D)
    monkeypatch.setenv("BTS_EDGES_SORTED", "1")
    exp = bts.DeviceTracker(max_tracks=K_tab, max_dets=M_cap, feat_dim=D)
    monkeypatch.delenv("BTS_EDGES_SORTED")
    ora = bo.TrackerOracle(faithful_iou=False) if oracle else None
    matched = 0
    for f in range(nframes):
        fr = frames_fn(f)
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        if ora is not None:
            ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            a, b = ref.stage_result(k), exp.stage_result(k)
            assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]), (f, k)
            if ora is not None:
                assert np.array_equal(b["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        matched += len(exp.stage_result(1)["m"])
        for which in (0, 1):
            ta, tb = ref.export(which), exp.export(which)
            for key in INT_KEYS:
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)
        # the sorted form leaves no candidate bit behind (tk_mid1 clears what the list names): the next frame starts clean
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, exp.counts())
        la, lb = ref.debug_lap(1), exp.debug_lap(1)
        assert int(la[0]) == int(lb[0]), (f, int(la[0]), int(lb[0]))          # the same number of stage-1 edges
    ref.close(); exp.close()
    return matched


@pytest.mark.parametrize("name", ["plain", "crowd@c2", "hard-reid@c2"])
def test_sorted_list_form_of_stage1_equals_the_tile_form(bts, monkeypatch, name):
    """BTS_EDGES_SORTED=1: stage-1 edges from tiles over spatially sorted columns (wave-union row skipping) plus candidate
    wavefronts fed by the pre-filter's candidate LIST.  Plain scene with churn, a dense crowd (staging-buffer overflow inside a
    tile, big components), clustered embeddings (more candidates per quadrant)."""
    synth = bts.synth
    if name == "plain":
        K, M, D = 600, 300, 256
        st = synth.SyntheticStream(K, M, D, seed=21, churn=0.05)
    else:
        K, M, D = synth.workload_dims(name)
        st = synth.make_workload(name, seed=77)
    matched = _sorted_vs_default(bts, monkeypatch, lambda f: st.next(), K + 256, max(K, M), D, 6)
    assert matched > 3 * M


def test_sorted_list_form_when_a_candidate_slot_overflows(bts, monkeypatch):
    """Every detection carries (nearly) the same feature row as every track: each 64 x 64 quadrant of the pre-filter holds far
    more candidates than a list slot (32) -- the consumer sweeps the quadrant's bitmap words instead, and tk_mid1 clears them."""
    rng = np.random.default_rng(8)
    K, D = 160, 128
    base = rng.standard_normal(D)
    base /= np.linalg.norm(base)

    class Fr:
        pass

    def frame(f):
        fr = Fr()
        cx = rng.permutation(K) * 40 + 10
        cy = rng.integers(100, 4000, K)
        fr.boxes = np.stack([cx, cy, cx + 30, cy + 60], 1).astype(np.int32)
        fr.scores = (rng.uniform(0.91, 0.99, K) if f == 0 else rng.uniform(0.5, 0.85, K)).astype(np.float32)
        noise = rng.standard_normal((K, D)) * 0.02
        fr.feats = (base[None, :] + noise).astype(np.float32)
        fr.feats /= np.linalg.norm(fr.feats, axis=1, keepdims=True)
        return fr

    matched = _sorted_vs_default(bts, monkeypatch, frame, 256, K, D, 4)
    assert matched > K


def test_sorted_list_form_at_a_size_that_is_one_pass_by_itself(bts, monkeypatch):
    """4224 x 2048 x 256 (528 pre-filter tiles > the 512 persistent workgroups): no test knob on the pre-filter, tiles of 32
    rows and more, cosines within +-0.01 of the gate with teleporting boxes (only candidates can re-associate)."""
    rng = np.random.default_rng(77)
    K, D = 2048, 256
    state = {"proto": None}
    proto = rng.standard_normal((K, D))
    state["proto"] = proto / np.linalg.norm(proto, axis=1, keepdims=True)

    class Fr:
        pass

    def frame(f):
        proto = state["proto"]
        fr = Fr()
        cx = rng.permutation(K) * 7 + 10
        cy = rng.integers(100, 8000, K)
        boxes = np.stack([cx, cy, cx + 5, cy + 9], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, K) if f == 0 else rng.uniform(0.5, 0.85, K)).astype(np.float32)
        c = 0.75 + rng.uniform(3e-5, 0.01, K) * rng.choice([-1.0, 1.0], K)
        if f == 0:
            c[:] = 1.0
        noise = rng.standard_normal((K, D))
        noise -= (noise * proto).sum(1, keepdims=True) * proto
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        feats = (c[:, None] * proto + np.sqrt(1 - c[:, None] ** 2) * noise).astype(np.float32)
        proto = feats.astype(np.float64) / np.linalg.norm(feats.astype(np.float64), axis=1, keepdims=True)
        perm = rng.permutation(K)
        fr.boxes, fr.scores, fr.feats = boxes[perm], scores[perm], np.ascontiguousarray(feats[perm])
        state["proto"] = proto[perm]
        return fr

    matched = _sorted_vs_default(bts, monkeypatch, frame, 4224, K, D, 3, oracle=False, onepass=False)
    assert matched > K // 2            # (frame 1: about half of the cosines inside the gate, frame 2: half of those)


@pytest.mark.parametrize("combo", ["ahead", "ahead+cvt", "ahead+sorted+cvt+predict_first", "ahead+predict_front", "ahead+predict_front+cvt+sorted",
                                   "merge_early", "ahead+merge_early+predict_front+cvt+sorted"])
@pytest.mark.parametrize("K,M,D", [(300, 150, 128), (2600, 700, 128)])
def test_lists_ahead_changes_no_result(bts, monkeypatch, combo, K, M, D):
    """BTS_LISTS_AHEAD=1: tk_finalize also builds the next frame's pool / unconfirmed / free lists (from tracked2 / lost2 and the
    drop flags, i.e. before the final lists exist) and the frame opens with tk_front_dets alone.  Alone and together with the
    other opt-in forms; with births, expiries and duplicate removal going on (churn), a camera-motion update between frames 1
    and 2 (it turns every float32 state into float64: the H_ALLF32 verdict taken ahead must follow) and a reset in the middle.
    The second size runs the split list jobs (chunks) where the table is large enough for them (BTS_CHUNK_ROWS)."""
    synth = bts.synth
    from oracle import bts_oracle as bo
    if K > 1000:
        monkeypatch.setenv("BTS_CHUNK_ROWS", "512")
    ref = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    if "ahead" in combo:
        monkeypatch.setenv("BTS_LISTS_AHEAD", "1")
    if "merge_early" in combo:                # (births + list merge in the launch that applies stages 2 and 3: tk_apply23_merge)
        monkeypatch.setenv("BTS_MERGE_EARLY", "1")
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    if "cvt" in combo:
        monkeypatch.setenv("BTS_CVT_OVERLAP", "1")
    if "sorted" in combo:
        monkeypatch.setenv("BTS_EDGES_SORTED", "1")
    if "predict_first" in combo:
        monkeypatch.setenv("BTS_PREDICT_FIRST", "1")
    if "predict_front" in combo:              # (the Kalman predict in the frame's first launch, none in the pre-filter's)
        monkeypatch.setenv("BTS_PREDICT_FRONT", "1")
    exp = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    assert exp.forms()["predict_front"] == ("1" if "predict_front" in combo else "0")
    assert exp.forms()["merge_early"] == ("1" if "merge_early" in combo else "0") and exp.forms()["lists_ahead"] == ("1" if "ahead" in combo else "0")
    for v in ("BTS_MERGE_EARLY", "BTS_LISTS_AHEAD", "BTS_CVT_OVERLAP", "BTS_EDGES_SORTED", "BTS_PREDICT_FIRST", "BTS_PREDICT_FRONT", "BTS_FILTER_ONEPASS"):
        monkeypatch.delenv(v, raising=False)
    exp.inputs_resident(True)
    ora = bo.TrackerOracle(faithful_iou=False)
    st = synth.SyntheticStream(K, M, D, seed=31, churn=0.08)
    H = np.array([[1.0, 0.002, 3.0], [-0.002, 1.0, -2.0]])
    nframes = 9 if K < 1000 else 5
    for f in range(nframes):
        if f == 1 and K < 1000:                  # camera motion between the first two frames (oracle: STrack.multi_gmc on both lists)
            ref.gmc(H); exp.gmc(H)
            bo.multi_gmc_lists(ora, H) if hasattr(bo, "multi_gmc_lists") else None
        if f == 6:
            ref.reset(); exp.reset()
            ora = bo.TrackerOracle(faithful_iou=False)
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        gmc_in_oracle = hasattr(bo, "multi_gmc_lists") or not (K < 1000 and 1 <= f < 6)
        if gmc_in_oracle:
            ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            a, b = ref.stage_result(k), exp.stage_result(k)
            assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]), (f, k)
            if gmc_in_oracle:
                assert np.array_equal(b["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        for which in (0, 1):
            ta, tb = ref.export(which), exp.export(which)
            for key in INT_KEYS + ("slot",):
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, ref.counts(), exp.counts())
    ref.close(); exp.close()


@pytest.mark.parametrize("K,M,D,F,combo", [(300, 150, 128, 0, "fused"), (200, 120, 128, 128, "fused"), (300, 150, 128, 0, "fused+merge_early+ahead"),
                                           (200, 120, 384, 256, "fused+planes"), (200, 120, 384, 0, "planes")])
def test_renorm_fused_changes_no_row(bts, monkeypatch, K, M, D, F, combo):
    """BTS_RENORM_FUSED=1: the in-place renormalisation (Q8, D:492-514) of a detection row that stage 1 matched moves from tk_mid1
    into the tk_mid2 workgroup that absorbs the row into its track.  Detection rows arrive with norms between 0.5 and 2, so the
    renormalisation is not a no-op: per frame the caller's feature array after the step (every row, matched or not), both
    feature tables of both lists, the assignments, the lists and the Kalman means must be bit-identical to the default form's."""
    import torch
    st = bts.synth.SyntheticStream(K, M, D, seed=77, churn=0.08, face_dim=F)
    rng = np.random.default_rng(5)
    ref = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D, face_dim=F)
    if "fused" in combo:
        monkeypatch.setenv("BTS_RENORM_FUSED", "1")
    if "planes" in combo:                     # (shadows in K-stage planes: conversion, copies into the table, births, the split-K pre-filter)
        monkeypatch.setenv("BTS_SHADOW_PLANES", "1")
    if "merge_early" in combo:
        monkeypatch.setenv("BTS_MERGE_EARLY", "1")
    if "ahead" in combo:
        monkeypatch.setenv("BTS_LISTS_AHEAD", "1")
    exp = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D, face_dim=F)
    assert exp.forms()["renorm_fused"] == ("1" if "fused" in combo else "0") and ref.forms()["renorm_fused"] == "0"
    assert exp.forms()["shadow_planes"] == ("1" if "planes" in combo else "0")
    for v in ("BTS_RENORM_FUSED", "BTS_MERGE_EARLY", "BTS_LISTS_AHEAD", "BTS_SHADOW_PLANES"):
        monkeypatch.delenv(v, raising=False)
    matched = 0
    for f in range(8):
        fr = st.next()
        scale = rng.uniform(0.5, 2.0, (len(fr.feats), 1)).astype(np.float32)
        feats = np.ascontiguousarray(fr.feats * scale)
        faces = np.ascontiguousarray(fr.face_feats * scale) if F else None
        outs = []
        for trk in (ref, exp):
            fa = torch.from_numpy(feats.copy()).cuda()
            fb = torch.from_numpy(faces.copy()).cuda() if F else None
            trk.step(torch.from_numpy(fr.boxes.astype(np.int32)).cuda(), torch.from_numpy(fr.scores.astype(np.float32)).cuda(), fa, fb)
            torch.cuda.synchronize()
            outs.append((fa.cpu().numpy(), fb.cpu().numpy() if F else None))
        assert np.array_equal(outs[0][0], outs[1][0]), f                     # the caller's rows (Q8), bit for bit
        if F:
            assert np.array_equal(outs[0][1], outs[1][1]), f
        if f > 0:
            assert not np.array_equal(outs[0][0], feats)                      # (and the renormalisation did something)
        for k in (1, 2, 3):
            a, b = ref.stage_result(k), exp.stage_result(k)
            assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]), (f, k)
        matched += len(ref.stage_result(1)["m"])
        for which in (0, 1):
            ta, tb = ref.export(which), exp.export(which)
            for key in INT_KEYS + ("slot",):
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)
            sa, sb = ref.export_state(which, cov=False, body=True, face=bool(F)), exp.export_state(which, cov=False, body=True, face=bool(F))
            for key in sa:
                assert np.array_equal(sa[key], sb[key]), (f, which, key)
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert matched > 3 * M
    ref.close(); exp.close()


@pytest.mark.parametrize("K,M,side,want", [(3400, 3200, 80.0, "B"), (4096, 1024, 40.0, "classic")])
def test_lap_self_form_on_dense_crowds(bts, monkeypatch, K, M, side, want):
    """BTS_LAP_SELF on dense crowds.  "B": 3400 x 3200 on a canvas that leaves ~5 000 non-trivial edges -- more than 4096, so the
    preparation takes its SECOND LDS layout, and the detections are wide enough (> 3072) for the instantiation that keeps 8 edges
    per thread in registers: every workgroup prepares the problem itself, ranks the components by their smallest row id and takes
    every (wavefronts of the launch)-th one.  "classic": 4096 x 1024, 5 300 edges but only 1024 columns -- the list is longer than
    what the 4-edges-per-thread instantiation keeps in registers, so the SELF form stands back and the launch runs the classic
    hand-over (every workgroup reaches that verdict from the edge count alone).  Assignments = the oracle's in both."""
    from oracle import bts_oracle as bo
    D = 64
    st = bts.synth.SyntheticStream(K, M, D, seed=77, side_scale=side)
    ora = bo.TrackerOracle(faithful_iou=False)
    monkeypatch.setenv("BTS_LAP_SELF", "1")
    trk = bts.DeviceTracker(max_tracks=K + 512, max_dets=max(K, M), feat_dim=D)
    monkeypatch.delenv("BTS_LAP_SELF")
    assert trk.forms()["lap_self"] == "1"
    most = 0
    for f in range(3):
        fr = st.next()
        trk.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            assert np.array_equal(trk.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        h = trk.debug_lap(1)
        if f >= 1:
            assert int(h[8]) == (3 if want == "B" else 1), (f, h[:10])
            most = max(most, int(h[6]))
    assert most > 4096, most                              # (non-trivial edges: beyond the first LDS layout)
    assert trk.counts()["errors"] == 0
    trk.close()


@pytest.mark.parametrize("combo", ["planes", "planes+ahead+merge_early+predict_front+sorted+self"])
def test_shadow_planes_one_pass_with_churn(bts, monkeypatch, combo):
    """BTS_SHADOW_PLANES through the persistent one-pass pre-filter (BTS_FILTER_ONEPASS: 4 x 2 tiles, 2 planes) on a stream with
    births, expiries and re-found tracks -- the shadows are written by the conversion (detections), copied into the table by the
    matches and the births, and read by the tiles in the plane layout; results = the default layout's and the oracle's."""
    from oracle import bts_oracle as bo
    K, M, D = 300, 150, 256
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    ref = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    monkeypatch.setenv("BTS_SHADOW_PLANES", "1")
    for key, env in (("ahead", "BTS_LISTS_AHEAD"), ("merge_early", "BTS_MERGE_EARLY"), ("predict_front", "BTS_PREDICT_FRONT"),
                     ("sorted", "BTS_EDGES_SORTED"), ("self", "BTS_LAP_SELF")):
        if key in combo:
            monkeypatch.setenv(env, "1")
    exp = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    assert exp.forms()["shadow_planes"] == "1" and ref.forms()["shadow_planes"] == "0"
    for v in ("BTS_FILTER_ONEPASS", "BTS_SHADOW_PLANES", "BTS_LISTS_AHEAD", "BTS_MERGE_EARLY", "BTS_PREDICT_FRONT", "BTS_EDGES_SORTED", "BTS_LAP_SELF"):
        monkeypatch.delenv(v, raising=False)
    ora = bo.TrackerOracle(faithful_iou=False)
    st = bts.synth.SyntheticStream(K, M, D, seed=13, churn=0.1)
    cands = 0
    for f in range(8):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        for k in (1, 2, 3):
            a, b = ref.stage_result(k), exp.stage_result(k)
            assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]), (f, k)
            assert np.array_equal(b["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        for which in (0, 1):
            ta, tb = ref.export(which), exp.export(which)
            for key in INT_KEYS + ("slot",):
                assert np.array_equal(ta[key], tb[key]), (f, which, key)
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)
        if f >= 1:
            sa, sb = ref.debug_filter(), exp.debug_filter()
            assert sa["candidates"] == sb["candidates"] and sa["tiles"] == sb["tiles"] and sa["k_stages"] == sb["k_stages"], (f, sa, sb)
            cands += sb["candidates"]
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert cands > 4 * M // 2
    ref.close(); exp.close()
