mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/t.log
tail -25 gpurun_out/t.log
