#!/bin/bash
mkdir -p gpurun_out/r2d
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r2d/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r2d/tests.log | tail -30
for wl in target c3 crowd c5; do
  timeout 900 python bench.py --workload $wl --no-cpu --no-export > gpurun_out/r2d/bench_$wl.json 2> gpurun_out/r2d/bench_$wl.err; echo "bench $wl rc=$?"
  tail -c 600 gpurun_out/r2d/bench_$wl.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2d/bench_$wl.json'))
    print('$wl', round(d['value'],1), 'steps/s', round(1e3*d['ms_per_step'],1),'us'); print(' roof', d['roofline']['kernel'][:24], round(d['roofline']['frac'],3), round(d['roofline']['avg_us'],1)); print(' filt', {k:v for k,v in d.get('prefilter').items() if k!='what'}); print(' seg', d['segments_us']); print(' lap', d['lap']['stage1'])
except Exception as e: print('$wl', 'parse failed', e)
PY
done
