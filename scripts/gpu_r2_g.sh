#!/bin/bash
# round-2 checkpoint g: whole GPU suite, rocprofv3 kernel statistics at the headline size, default bench line
mkdir -p gpurun_out/r2g
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r2g/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|ERROR" gpurun_out/r2g/tests.log | tail -12
(cd /tmp && export TMPDIR=/tmp && for w in target c3; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/prof_$w -o $w -- python $R/bench.py --workload $w --steps 100 --no-cpu --no-export > $R/gpurun_out/r2g/prof_$w.log 2>&1
  (cd $R && python scripts/prof_top.py gpurun_out/r2g/prof_$w > gpurun_out/r2g/${w}_kernel_stats.csv; echo "== $w"; head -16 gpurun_out/r2g/${w}_kernel_stats.csv; rm -rf gpurun_out/r2g/prof_$w)
done)
timeout 1200 python bench.py > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err; echo "bench rc=$?"
tail -c 500 gpurun_out/r2g/bench_default.err | grep -v amdgpu
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2g/bench_default.json'))
print(round(d['value'],1),'steps/s', d['config']['workload'][:40]); print(d['roofline']['step']); print({k:d['roofline'][k] for k in ('kernel','frac','avg_us')}); print(d['export_inclusive']['async_steps_per_s'], d['export_inclusive']['sync_steps_per_s']); print(d['cpu_baseline']['value'], d['id_match']['frames_identical'], d['id_match']['frames'])
PY
