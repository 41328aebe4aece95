#!/bin/bash
# round-2 checkpoint b: int8 pre-filter with prefix exits -- whole GPU suite, then bench lines
mkdir -p gpurun_out/r2b
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r2b/tests.log
for wl in target c3 c5 c2 hard-reid crowd; do
  timeout 900 python bench.py --workload $wl --no-cpu --no-export > gpurun_out/r2b/bench_$wl.json 2> gpurun_out/r2b/bench_$wl.err; echo "bench $wl rc=$?"
  tail -c 400 gpurun_out/r2b/bench_$wl.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2b/bench_$wl.json'))
    print('$wl', round(d['value'],1), 'steps/s', round(1e3*d['ms_per_step'],1),'us'); print(' roof', d['roofline']['kernel'][:24], round(d['roofline']['frac'],3), round(d['roofline']['avg_us'],1)); print(' filt', d.get('prefilter')); print(' seg', d['segments_us']); print(' lap', d['lap']['stage1'])
except Exception as e: print('$wl', 'parse failed', e)
PY
done
