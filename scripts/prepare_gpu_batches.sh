#!/bin/bash
# Build what the GPU batches need and git does not carry (ab/ is git-ignored but travels with gpurun): the trace variants of HEAD's
# sources.  Run on the CPU box before `gpurun -- 'bash scripts/gpu_r5_a.sh'` / gpu_r5_b.sh.
set -e
cd "$(dirname "$0")/.."
make -C bot-sort-onnx-tensorrt_amd/csrc -j8 > /dev/null
bash scripts/build_variant.sh ftrace FTRACE=1
bash scripts/build_variant.sh trace TRACE=1
git rev-parse --short HEAD > gpurun_out/commit.txt 2>/dev/null || true
