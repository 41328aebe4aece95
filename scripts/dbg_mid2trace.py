"""Role timeline of tk_mid2 (library built with make TRACE=1): latest end of the Kalman / feature / stage-2 / stage-3
workgroups relative to the first workgroup's start."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
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
        if f >= 8: acc.append(((h[33:37] - h[32]).astype(np.int64) & 0xffffffff).astype(np.float64) / 100.0)
    a = np.mean(acc, axis=0)
    print(wl, "tk_mid2: Kalman done %.1f us | feature rows done %.1f | stage-2 tiles done %.1f | stage-3 tiles done %.1f" % tuple(a))
    trk.close()
