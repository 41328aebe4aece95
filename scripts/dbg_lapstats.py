import importlib, sys, numpy as np, torch, ctypes
sys.path.insert(0, '.')
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
for wl in sys.argv[1:]:
    K, M, D = pkg.synth.CONFIGS[wl]
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    for f in range(8):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
    for stage in (1, 2, 3):
        h = np.zeros(64, dtype=np.int32)
        pkg._lib.check(pkg._lib.load().bts_tracker_debug_lap(trk.handle, stage, h.ctypes.data, pkg._lib.current_stream()))

        print(wl, "stage", stage,  "E=%d comps=%d hard_edges=%d mode=%d ready=%d" % (h[0], h[1], h[6], h[8], h[7]))
    trk.close()
