"""Steps/s when every frame's arrays start in (pinned) HOST memory: the upload of boxes / scores / features rides the
step's stream in front of the 11 launches.  (The C ABI takes device pointers; bench.py's `value` has the frames
resident in HBM.  This is the PCIe-inclusive figure DESIGN.md quotes beside it.)  usage: pcie_rate.py c2 target ..."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
STEPS, WARM = 200, 10
for wl in sys.argv[1:] or ["c2"]:
    K, M, D = pkg.synth.CONFIGS[wl]
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234)
    torch.cuda.set_stream(torch.cuda.Stream())
    trk = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
    host = []
    for _ in range(WARM + STEPS):
        fr = st.next()
        host.append(tuple(torch.from_numpy(a).pin_memory() for a in (fr.boxes, fr.scores, fr.feats)))
    big = max(host, key=lambda h: h[0].shape[0])
    dev = [torch.empty_like(a, device="cuda") for a in big]          # one landing buffer per array, reused every frame

    def run(frames, upload):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h in frames:
            if upload:
                n = h[0].shape[0]
                for d, s in zip(dev, h):
                    d[:n].copy_(s, non_blocking=True)
                trk.step(*(d[:n] for d in dev))
            else:
                trk.step(*h)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(host[:WARM], True)
    t = run(host[WARM:], True)
    mb = sum(a.numel() * a.element_size() for a in host[-1]) / 1e6
    print("%s: %.0f steps/s with the per-frame upload (%.2f MB/frame from pinned host memory, %.1f us/step, %.1f GB/s of PCIe payload)"
          % (wl, STEPS / t, mb, 1e6 * t / STEPS, mb * STEPS / t / 1e3), flush=True)
    assert trk.counts()["errors"] == 0
    trk.close()
