#!/bin/bash
mkdir -p gpurun_out/r2e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_round2.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r2e/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r2e/tests.log | tail -10
for wl in target c3 hard-reid; do
  timeout 900 python bench.py --workload $wl --no-cpu --no-export > gpurun_out/r2e/bench_$wl.json 2> gpurun_out/r2e/bench_$wl.err; echo "bench $wl rc=$?"
  tail -c 700 gpurun_out/r2e/bench_$wl.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2e/bench_$wl.json'))
    print('$wl', round(d['value'],1), 'steps/s', round(1e3*d['ms_per_step'],1),'us'); print(' roof', d['roofline']['kernel'][:24], round(d['roofline']['frac'],3), round(d['roofline']['avg_us'],1)); print(' filt', {k:v for k,v in d.get('prefilter').items() if k!='what'}); print(' seg', d['segments_us']); print(' lap', d['lap']['stage1'])
except Exception as e: print('$wl', 'parse failed', e)
PY
done
