#!/bin/bash
mkdir -p gpurun_out/r2f
for wl in c3 target; do
  BTS_BENCH_IGNORE_ERRORS=1 timeout 900 python bench.py --workload $wl --no-cpu --no-export --steps 60 > gpurun_out/r2f/bench_${wl}.json 2> gpurun_out/r2f/bench_${wl}.err; echo "bench $wl rc=$?"
  tail -c 300 gpurun_out/r2f/bench_${wl}.err | grep -v amdgpu
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2f/bench_${wl}.json'))
    f=d['prefilter']; tot=f['cycles_total']
    print(' k_filter us', round(d['roofline']['avg_us'],1), 'stages', f['k_stages'], 'tiles', f['tiles'])
    nwg=min(512,f['tiles']); u=lambda x: round(x*0.01/nwg,2)
    print('  per-WG averages (us, 100 MHz clock): total', u(tot/64), 'prologue', u(f['tests']), 'first test', u(f['candidates']), 'other tests', u(f['cycles_tests']/64), 'stage waits', u(f['cycles_stage_waits']/64), ' tiles/WG', round(f['tiles']/nwg,2), 'stages/WG', round(f['k_stages']/nwg,1))
    r=d['prefilter'].get('raw')
    if r: print('  GEMM WGs: first start 0, last start %.2f us, last end %.2f us, longest WG %.2f us; predict WGs: first start %.2f, last end %.2f us' % ((r[11]-r[8])*0.01, (r[9]-r[8])*0.01, r[10]*0.01, (r[12]-r[8])*0.01, (r[13]-r[8])*0.01))
except Exception as e: print('parse failed', e)
PY
done
