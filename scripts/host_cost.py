"""Host-side enqueue cost of bts_tracker_step vs the GPU time of a step (is the chain launch-bound on the host?)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
w = sys.argv[1] if len(sys.argv) > 1 else "c2"
K, M, D = pkg.synth.CONFIGS[w]
st = pkg.synth.SyntheticStream(K, M, D, seed=1)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
def nf():
    fr = st.next(); return (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
fr = [nf() for _ in range(60)]
for f in fr[:20]: trk.step(*f)
torch.cuda.synchronize()
for n in (5, 10, 20, 40):
    t0 = time.perf_counter()
    for f in fr[20:20 + n]: trk.step(*f)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(w, "n=%d enqueue %.1f us/step, to-sync total %.1f us/step" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
