"""Full table of the standalone kernels against the MI355X rooflines (see bot-sort-onnx-tensorrt_amd/microbench.py)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mb = importlib.import_module("bot-sort-onnx-tensorrt_amd.microbench")
for r in mb.run(full=True):
    print("%-18s %-22s %8.1f us  %8.1f %-7s (%.0f%% of %s)  %d operand sets, %d MB rotated%s" % (
        r["kernel"], r["shape"], r["avg_us"], r["achieved"], r["unit"], 100 * r["frac"],
        "8 TB/s HBM" if r["unit"] == "GB/s" else "157 TF f32 MFMA", r["operand_sets"], r["rotated_MB"],
        "  [" + r["note"] + "]" if "note" in r else ""))
