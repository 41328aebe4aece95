#!/bin/bash
mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_sanitizers.py tests/test_gpu_round2.py -q -m gpu -p no:cacheprovider -k "asan or neighbour" 2>&1 | tail -15
timeout 900 python scripts/micro_kernels.py > gpurun_out/r2l/micro_kernels.txt 2>&1; cat gpurun_out/r2l/micro_kernels.txt | grep -v amdgpu.ids
