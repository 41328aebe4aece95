import importlib, sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
lib = pkg._lib.load()
lib.bts_tracker_debug_stamps.restype = ctypes.c_int
lib.bts_tracker_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for wl in sys.argv[1:]:
    K, M, D = pkg.synth.CONFIGS[wl]
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    acc = np.zeros(18); n = 0
    for f in range(14):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
        buf = np.zeros(32, dtype=np.uint64)
        lib.bts_tracker_debug_stamps(trk.handle, buf.ctypes.data)
        if f >= 4:
            b = buf.astype(np.int64)[:18]
            acc += (b - b[0]) / 100.0
            n += 1
    print(wl, "us since kernel-entry stamp:", " ".join("%d:%.1f" % (k, v / n) for k, v in enumerate(acc)))
    trk.close()
