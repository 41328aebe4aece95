import importlib, sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
lib = pkg._lib.load()
lib.bts_tracker_debug_stamps.restype = ctypes.c_int
lib.bts_tracker_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
names = ["front.pool", "front.unc", "front.dets", "front.free", "front.cvt(first..last)", "mid1.r2", "mid1.left", "mid1.uncbox", "mid1.renorm(first..last)",
         "merge.births", "merge.keep", "merge.refind", "merge.lost0", "merge.lost1"]
for wl in sys.argv[1:]:
    K, M, D = pkg.synth.CONFIGS[wl]
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    acc = np.zeros(len(names)); n = 0
    for f in range(14):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
        buf = np.zeros(128, dtype=np.uint64)
        lib.bts_tracker_debug_stamps(trk.handle, buf.ctypes.data)
        if f >= 4:
            b = buf.astype(np.int64).reshape(16, 8)
            acc += (b[:len(names), 1] - b[:len(names), 0]) / 100.0      # wall_clock64: 100 MHz -> us
            n += 1
    print(wl, " ".join("%s=%.1f" % (nm, v / n) for nm, v in zip(names, acc)))
    trk.close()
