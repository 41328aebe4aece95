"""Is the several-streams-per-GPU mode bound by the host (enqueue) or by the device?  Prints, for S threads, the time
the slowest thread needed to ENQUEUE its steps and the time until the device drained."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
K, M, D = pkg.synth.CONFIGS["c2"]
N = 400
DISTINCT = len(sys.argv) > 1
for S in (1, 4):
    trk, frs, hs = [], [], []
    for k in range(S):
        st = pkg.synth.SyntheticStream(K, M, D, seed=1234 + k)
        t = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
        f = []
        for i in range(4 + (N if DISTINCT else 8)):
            fr = st.next()
            x = (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
            if i < 4: t.step(*x)
            else: f.append(x)
        trk.append(t); frs.append(f); hs.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    done = [0.0] * S
    gate = threading.Barrier(S + 1)
    def worker(k):
        with torch.cuda.stream(hs[k]):
            gate.wait()
            for i in range(N):
                trk[k].step(*frs[k][i if DISTINCT else i % 8])
            done[k] = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
    for t in th: t.start()
    t0 = time.perf_counter(); gate.wait()
    for t in th: t.join()
    t_enq = max(done) - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("S=%d: enqueue done after %.1f ms (%.1f us/step/thread), device drained after %.1f ms -> %.0f steps/s" % (S, 1e3 * t_enq, 1e6 * t_enq / N, 1e3 * t_all, S * N / t_all))
    for t in trk: t.close()
ms = importlib.import_module("bot-sort-onnx-tensorrt_amd.multistream")
for S in (4,):
    trk, frs, hs = [], [], []
    for k in range(S):
        st = pkg.synth.SyntheticStream(K, M, D, seed=1234 + k)
        t = pkg.DeviceTracker(max_tracks=K + 64, max_dets=max(K, M), feat_dim=D)
        f = []
        for i in range(4 + N):
            fr = st.next()
            x = (torch.from_numpy(fr.boxes).cuda(), torch.from_numpy(fr.scores).cuda(), torch.from_numpy(fr.feats).cuda())
            if i < 4: t.step(*x)
            else: f.append(x)
        trk.append(t); frs.append(f); hs.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    el = ms.run_concurrent(trk, frs, hs)
    print("run_concurrent S=%d: %.0f steps/s" % (S, S * N / el))
    main = torch.cuda.Stream(); torch.cuda.set_stream(main)
    el = ms.run_concurrent(trk, frs, [main] + hs[1:])
    print("run_concurrent with the main thread's current stream as stream 0: %.0f steps/s" % (S * N / el))
