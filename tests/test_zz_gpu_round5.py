"""(File name: sorts after the suites that have been green on hardware, like the other test_zz_gpu_* files.)

Round-5 GPU parity tests -- written while GPU access was still closed and run on the CPU emulation of the kernel sources
(tests/hip_emu):
  * the shape-generic one-pass pre-filter (k_filter_t, BTS_FILTER_TILE=160x128 | 128x64): FILTER = DENSE on cosines within
    +-0.01 of the gate at 384 x 256 x 2048; identical results to the 128 x 128 tiles at 4224 x 2048 x 256 (more tiles than resident
    workgroups: next-tile prefetch, stop prediction, ragged last tiles in both directions); a stream with births / expiries /
    re-found tracks against the oracle; combined with the other forms;
  * boolean forms given as 0 / false switch a form OFF on a live tracker (bts_tracker_forms)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_KEYS = ("track_id", "state", "is_activated", "frame_id", "start_frame", "tracklet_len")
TILES = {"160x128": (160, 128), "128x64": (128, 64)}


@pytest.fixture(scope="module")
def bts():
    pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
    pkg._lib.load(require_gpu=True)
    return pkg


def _dev(fr):
    import torch
    return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats.copy()).cuda())


class _GateBand:
    """Frames whose cosines to the tracks' rows sit within +-0.01 of the appearance gate while the boxes teleport: only pairs the
    pre-filter hands to the exact refinement can re-associate."""

    def __init__(self, K, D, seed, inside=None):
        self.rng = np.random.default_rng(seed)
        self.K, self.D = K, D
        self.inside = inside                 # (lo, hi): every cosine inside the gate instead (the stream then sustains itself)
        proto = self.rng.standard_normal((K, D))
        self.proto = proto / np.linalg.norm(proto, axis=1, keepdims=True)
        self.f = 0

    def next(self):
        rng, K, D, proto = self.rng, self.K, self.D, self.proto
        cx = rng.permutation(K) * 7 + 10
        cy = rng.integers(100, 8000, K)
        boxes = np.stack([cx, cy, cx + 5, cy + 9], 1).astype(np.int32)
        scores = (rng.uniform(0.91, 0.99, K) if self.f == 0 else rng.uniform(0.5, 0.85, K)).astype(np.float32)
        c = 0.75 + rng.uniform(3e-5, 0.01, K) * rng.choice([-1.0, 1.0], K)
        if self.inside:
            c = rng.uniform(self.inside[0], self.inside[1], K)
        if self.f == 0:
            c[:] = 1.0
        noise = rng.standard_normal((K, D))
        noise -= (noise * proto).sum(1, keepdims=True) * proto
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        feats = (c[:, None] * proto + np.sqrt(1 - c[:, None] ** 2) * noise).astype(np.float32)
        proto = feats.astype(np.float64) / np.linalg.norm(feats.astype(np.float64), axis=1, keepdims=True)
        perm = rng.permutation(K)

        class Fr:
            pass
        fr = Fr()
        fr.boxes, fr.scores, fr.feats = boxes[perm], scores[perm], np.ascontiguousarray(feats[perm])
        self.proto = proto[perm]
        self.f += 1
        return fr


def _same_frame(a, b, f, means=True):
    for k in (1, 2, 3):
        ra, rb = a.stage_result(k), b.stage_result(k)
        assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["y"], rb["y"]), (f, k)
    for which in (0, 1):
        ta, tb = a.export(which), b.export(which)
        for key in INT_KEYS:
            assert np.array_equal(ta[key], tb[key]), (f, which, key)
        if means:
            assert np.array_equal(ta["mean"], tb["mean"]), (f, which)


@pytest.mark.parametrize("extra", ["", "guess", "planes", "first16_8", "predict_first"])
@pytest.mark.parametrize("tile", ["160x128", "128x64"])
def test_prefilter_tile_shapes_vs_dense(bts, monkeypatch, tile, extra):
    """k_filter_t at 384 x 256 x 2048 (BTS_FILTER_ONEPASS sends the problem through the one-pass form; 256 live rows and columns):
    2 x 2 live tiles of 160 x 128 -- the last row tile holds 96 live rows -- or 2 x 4 of 128 x 64; 16 K stages, checkpoints in use.  FILTER with
    that tile shape must equal DENSE frame by frame; also with the first-tile stop prediction (two persistent workgroups: the
    next-tile prefetch paths), the plane layout of the shadows and a late first checkpoint."""
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    monkeypatch.setenv("BTS_FILTER_TILE", tile)
    if extra == "guess":
        monkeypatch.setenv("BTS_FILTER_GUESS", "1")
        monkeypatch.setenv("BTS_FILTER_PW", "2")
    elif extra == "planes":
        monkeypatch.setenv("BTS_SHADOW_PLANES", "1")
    elif extra == "first16_8":
        monkeypatch.setenv("BTS_I8_FIRST16", "8")
    elif extra == "predict_first":                      # the Kalman-predict workgroups in FRONT of the tile workgroups (a kernel argument)
        monkeypatch.setenv("BTS_PREDICT_FIRST", "1")
    K, D = 256, 2048
    a = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D)
    assert a.forms()["filter_tile"] == tile
    b = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D, appearance_mode=1)
    src = _GateBand(K, D, 41)
    early, matched1 = 0, 0
    tm, tn = TILES[tile]
    for f in range(4):
        fr = src.next()
        a.step(*_dev(fr)); b.step(*_dev(fr))
        _same_frame(a, b, f, means=False)
        if f == 1:
            matched1 = len(a.stage_result(1)["m"])
        if f >= 1:
            st = a.debug_filter()
            assert st["tile"] == (tm, tn)
            assert st["tiles"] == -(-K // tm) * -(-K // tn) and st["k_stages"] >= st["tiles"] * 3, st
            early += st["prefix_exits"]
            assert st["candidates"] >= len(a.stage_result(1)["m"])          # every appearance match was a candidate
    assert a.counts() == b.counts() and a.counts()["errors"] == 0
    assert matched1 > K // 4 and early > 0
    a.close(); b.close()


@pytest.mark.parametrize("tile", ["160x128", "128x64"])
def test_prefilter_tile_shapes_at_a_one_pass_size(bts, monkeypatch, tile):
    """4224 x 2048 x 256, no test knob: 27 x 16 = 432 tiles of 160 x 128 (one per workgroup) or 33 x 32 = 1 056 of 128 x 64 over
    768 persistent workgroups (next-tile prefetch, stop prediction over several tiles); cosines within +-0.01 of the gate.  The
    tracker with the new tile shape must return what the 128 x 128 tiles return, frame by frame, and the same number of stage-1
    edges (the candidate SETS may differ -- both are supersets of what the exact refinement accepts)."""
    K, D = 2048, 256
    ref = bts.DeviceTracker(max_tracks=4224, max_dets=K, feat_dim=D)
    monkeypatch.setenv("BTS_FILTER_TILE", tile)
    exp = bts.DeviceTracker(max_tracks=4224, max_dets=K, feat_dim=D)
    monkeypatch.delenv("BTS_FILTER_TILE")
    assert exp.forms()["filter_tile"] == tile and ref.forms()["filter_tile"] == "128x128"
    src = _GateBand(K, D, 77)
    matched = 0
    for f in range(3):
        fr = src.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        _same_frame(ref, exp, f)
        matched += len(exp.stage_result(1)["m"])
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0, (f, exp.counts())
        assert int(ref.debug_lap(1)[0]) == int(exp.debug_lap(1)[0]), f
        if f >= 1:
            sa, sb = ref.debug_filter(), exp.debug_filter()
            assert sa["tile"] == (128, 128) and sb["tile"] == TILES[tile]
            assert sb["tiles"] == -(-K // TILES[tile][0]) * -(-K // TILES[tile][1]), sb        # live tiles: pool rows x high-score columns
            assert sb["candidates"] >= len(exp.stage_result(1)["m"])
    assert matched > K // 2
    ref.close(); exp.close()


@pytest.mark.parametrize("combo", ["", "ahead+merge_early+predict_front+renorm_fused+self"])
@pytest.mark.parametrize("tile", ["160x128", "128x64"])
def test_prefilter_tile_shapes_with_churn_vs_oracle(bts, monkeypatch, tile, combo):
    """A stream with births, expiries and re-found tracks (300 identities, 150 detections per frame, 10 % churn) through the
    one-pass form with the new tile shapes: ragged extents in both directions (pool rows and high-score columns are no multiples of
    the tile), rows past the live extents inside edge tiles; results = the default tiles' and the oracle's, slots included."""
    from oracle import bts_oracle as bo
    K, M, D = 300, 150, 256
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    ref = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    monkeypatch.setenv("BTS_FILTER_TILE", tile)
    envs = {"ahead": "BTS_LISTS_AHEAD", "merge_early": "BTS_MERGE_EARLY", "predict_front": "BTS_PREDICT_FRONT", "renorm_fused": "BTS_RENORM_FUSED", "self": "BTS_LAP_SELF"}
    for key in [c for c in combo.split("+") if c]:
        monkeypatch.setenv(envs[key], "1")
    exp = bts.DeviceTracker(max_tracks=K + K // 2, max_dets=max(K, M), feat_dim=D)
    assert exp.forms()["filter_tile"] == tile
    for v in ["BTS_FILTER_ONEPASS", "BTS_FILTER_TILE"] + list(envs.values()):
        monkeypatch.delenv(v, raising=False)
    ora = bo.TrackerOracle(faithful_iou=False)
    st = bts.synth.SyntheticStream(K, M, D, seed=13, churn=0.1)
    cands = 0
    for f in range(8):
        fr = st.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        ora.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats.copy(), None)
        _same_frame(ref, exp, f)
        for k in (1, 2, 3):
            assert np.array_equal(exp.stage_result(k)["m"], np.asarray(ora.trace["m%d" % k]).reshape(-1, 2)), (f, k)
        for which in (0, 1):
            assert np.array_equal(ref.export(which)["slot"], exp.export(which)["slot"]), (f, which)
        if f >= 1:
            cands += exp.debug_filter()["candidates"]
        assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert cands > 4 * M // 2
    ref.close(); exp.close()


def test_a_boolean_form_given_as_zero_is_off(bts):
    """DeviceTracker(forms={'BTS_LISTS_AHEAD': 0}) used to ENABLE the form ("KEY=0" is non-empty)."""
    for forms, want in (({"BTS_LISTS_AHEAD": 0, "BTS_MERGE_EARLY": False}, "0"), ("BTS_LISTS_AHEAD=0,BTS_MERGE_EARLY=off", "0"),
                        ({"BTS_LISTS_AHEAD": 1, "BTS_MERGE_EARLY": True}, "1")):
        t = bts.DeviceTracker(max_tracks=256, max_dets=64, feat_dim=128, forms=forms)
        eff = t.forms()
        t.close()
        assert eff["lists_ahead"] == want and eff["merge_early"] == want, (forms, eff)


def test_template_at_the_default_shape_is_k_filter_i8(bts, monkeypatch):
    """Emulator build only (the product library does not carry this instantiation): k_filter_t<128, 128, 2 x 2, two record images>
    is k_filter_i8 written through the shape parameters -- same tiles, same tests, same stop decisions -- so the two kernels must
    agree on every COUNTER of every frame (tiles, K stages run, prefix exits, candidates), not only on the step's results.  What
    the generalisation could have broken (fragment map, record staging, row summary, live-extent masks, emit indices) shows here."""
    monkeypatch.setenv("BTS_FILTER_ONEPASS", "1")
    K, D = 256, 1024
    ref = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D)
    monkeypatch.setenv("BTS_FILTER_TILE", "128x128t")
    exp = bts.DeviceTracker(max_tracks=384, max_dets=K, feat_dim=D)
    if exp.forms().get("filter_tile") != "128x128t":
        ref.close(); exp.close()
        pytest.skip("only the CPU emulation build instantiates the template at 128 x 128")
    src = _GateBand(K, D, 5)
    for f in range(4):
        fr = src.next()
        ref.step(*_dev(fr)); exp.step(*_dev(fr))
        _same_frame(ref, exp, f)
        if f >= 1:
            sa, sb = ref.debug_filter(), exp.debug_filter()
            for key in ("tiles", "k_stages", "prefix_exits", "candidates", "tile"):
                assert sa[key] == sb[key], (f, key, sa, sb)
    assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    ref.close(); exp.close()


@pytest.mark.parametrize("ahead", [1, 0])
@pytest.mark.parametrize("resident", [True, False])
def test_cvt_overlap_without_host_synchronisation_between_steps(bts, monkeypatch, resident, ahead):
    """BTS_CVT_OVERLAP (+ BTS_LISTS_AHEAD, the pairing it is offered in) over 24 frames whose boxes teleport while the cosines stay
    inside the gate -- ONLY the appearance candidates, i.e. the converted detection shadows, can associate a frame, so a conversion
    that ran at the wrong time changes every assignment -- digests compared with the default forms' every SIXTH frame only -- nothing synchronises the host with the device in between, so on a GPU the side
    stream's conversion of frame t + 1 really runs under frame t's kernels.  On the CPU emulation this test is also run under
    HIPEMU_STREAMS=sidelate / mainlate (tests/test_emu_gpu_subset.py): one of the two streams then lags as far behind the other as
    the form's events allow, and a dependency the form forgot to state changes a digest."""
    tune = importlib.import_module("bot-sort-onnx-tensorrt_amd.tune")
    K, M, D = 96, 96, 128
    ref = bts.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    exp = bts.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D, forms={"BTS_CVT_OVERLAP": 1, "BTS_LISTS_AHEAD": ahead})
    assert exp.forms()["cvt_overlap"] == "1" and ref.forms()["cvt_overlap"] == "0" and exp.forms()["lists_ahead"] == str(ahead)
    exp.inputs_resident(resident)
    st = _GateBand(K, D, 31, inside=(0.8, 0.92))
    frames = [_dev(st.next()) for _ in range(24)]
    frames2 = [(b, s, f.clone()) for (b, s, f) in frames]
    import torch
    torch.cuda.synchronize()
    for f, (fa, fb) in enumerate(zip(frames, frames2)):
        ref.step(*fa)
        exp.step(*fb)
        if f % 6 == 5:
            assert tune._digest(ref) == tune._digest(exp), f
    assert tune._state_digest(ref) == tune._state_digest(exp)
    assert ref.counts() == exp.counts() and exp.counts()["errors"] == 0
    assert ref.counts()["n_tracked"] > K // 2           # (the stream sustained itself on appearance alone)
    ref.close(); exp.close()
