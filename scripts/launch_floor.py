"""What a dependent launch costs on this GPU: empty kernels and kernels that follow a chain of dependent loads."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
lib = pkg._lib.load(require_gpu=True); ctx = pkg._lib.default_context().handle
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
buf = torch.zeros(16384, dtype=torch.int32, device="cuda")
def run(blocks, threads, links, n=200):
    f = lambda k: pkg._lib.check(lib.bts_debug_launch_floor(ctx, pkg._lib.ptr(buf), k, blocks, threads, links, pkg._lib.current_stream()))
    f(20); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(n); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for blocks, threads in ((1, 64), (1, 1024), (16, 1024), (256, 256), (1024, 256)):
    print("%5d x %4d threads: " % (blocks, threads) + "  ".join("%d links %.2f us" % (l, run(blocks, threads, l)) for l in (0, 1, 4, 16, 64)))
