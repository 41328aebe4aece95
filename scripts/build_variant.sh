#!/bin/bash
# Build a VARIANT of the library into ab/<name>.so without touching the package's objects: the sources are copied to a scratch
# directory and built there with extra make variables.  ab/ is git-ignored but travels to the GPU box (gpurun).
#   scripts/build_variant.sh ftrace FTRACE=1          # k_filter_i8 with cycle accounting (scripts/dbg_filtertrace.py)
#   scripts/build_variant.sh trace TRACE=1            # k_lap / list-job phase stamps
#   scripts/build_variant.sh head                     # plain copy of the current sources (the A side of an A/B)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/btsvar.XXXXXX)
mkdir -p "$W/pkg/csrc" "$W/include" "$R/ab"
cp "$R"/bot-sort-onnx-tensorrt_amd/csrc/*.hip "$R"/bot-sort-onnx-tensorrt_amd/csrc/*.h "$R"/bot-sort-onnx-tensorrt_amd/csrc/Makefile "$W/pkg/csrc/"
cp "$R"/include/*.h "$W/include/"
# (the Makefile refers to ../../include: keep that shape)
mkdir -p "$W/top/pkg" && mv "$W/pkg/csrc" "$W/top/pkg/csrc" && mv "$W/include" "$W/top/include"
make -C "$W/top/pkg/csrc" -j8 OUT="$R/ab/$name.so" "$@" > "$W/build.log" 2>&1 || { tail -20 "$W/build.log"; exit 1; }
rm -rf "$W"
ls -la "$R/ab/$name.so"
