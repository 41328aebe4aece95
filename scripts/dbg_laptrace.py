import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
for wl in sys.argv[1:]:
    K, M, D = pkg.synth.workload_dims(wl)
    st = pkg.synth.make_workload(wl, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=2 * K, max_dets=max(K, M), feat_dim=D)
    acc = []
    for f in range(40):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
        h = np.zeros(64, dtype=np.int32)
        pkg._lib.check(pkg._lib.load().bts_tracker_debug_lap(trk.handle, 1, h.ctypes.data, pkg._lib.current_stream()))
        if f >= 8: acc.append((h[9:9 + 17] - h[9]).astype(np.int64) & 0xffffffff)
    a = np.mean(np.array(acc, dtype=np.float64), axis=0) / 100.0
    names = ["entry", "E loaded", "classified", "slots", "roots", "fits", "grouped", "exported", "published", "helper entry", "helper ready", "prep WG done", "helper WG done", "h: solve entry", "h: cdesc", "h: loaded+ranked", "h: solved"]
    print(wl, " | ".join("%s %.2f" % (n, v) for n, v in zip(names, a)))
    trk.close()
