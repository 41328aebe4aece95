"""Durations of the single-workgroup list jobs inside tk_front / tk_mid1 / tk_merge / tk_finalize (library built with make TRACE=1)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
names = {0: ["prep_pool", "prep_unconfirmed", "prep_dets", "prep_free", "cvt(first)"], 1: ["r2 rows", "left dets", "unc boxes", "renorm(first)"],
         2: ["births", "keep", "re-found", "old lost+expiry", "newly lost"], 3: ["tracked", "lost"]}
for wl in sys.argv[1:]:
    K, M, D = pkg.synth.workload_dims(wl)
    st = pkg.synth.make_workload(wl, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    acc = []
    for f in range(24):
        fr = st.next()
        trk.step(torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
        torch.cuda.synchronize()
        h = np.zeros(64, dtype=np.int32)
        pkg._lib.check(pkg._lib.load().bts_debug_jobtrace(h.ctypes.data))
        if f >= 8: acc.append(h.astype(np.float64))
    a = np.mean(acc, axis=0) / 100.0
    for k, kn in enumerate(("tk_front", "tk_mid1", "tk_merge", "tk_finalize")):
        print(wl, kn, "  ".join("%s %.1f" % (n, a[k * 8 + j]) for j, n in enumerate(names[k])))
    trk.close()
