#!/bin/bash
# Verification of branch `tall-noaccum` (split-K form compiled out of the tall stage-1 cost tiles: 0 B of scratch again).
# Run from a checkout of that branch after `python -c "import __graft_entry__ as g; g.build()"`, with the main build kept as
# ab/head.so:   gpurun -- 'bash scripts/gpu_tall_noaccum.sh'
#   1. the whole GPU suite on the branch build (parity first);
#   2. A/B of the edges kernel and the step rate, interleaved on one box, at three sizes.
mkdir -p gpurun_out/tall
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tall/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/tall/tests.log | tail -6
cp bot-sort-onnx-tensorrt_amd/libbotsort_hip.so ab/tall_noaccum.so
for w in target c3 c5; do echo "== $w"; VARIANTS="head tall_noaccum" WL=$w KERN=tk_edges bash scripts/gpu_ab2.sh; done
