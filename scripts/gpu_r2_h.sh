#!/bin/bash
mkdir -p gpurun_out/r2h
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r2h/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r2h/tests.log | tail -5
cd /tmp && export TMPDIR=/tmp
for w in target c3 hard-reid; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2h/prof_$w -o p -- python $R/bench.py --workload $w --steps 60 --no-cpu --no-export > $R/gpurun_out/r2h/prof_$w.log 2> $R/gpurun_out/r2h/prof_$w.err
  (cd $R && python scripts/prof_top.py gpurun_out/r2h/prof_$w | grep -E "k_filter|tk_edges" | head -2 | sed "s/^/$w /"; rm -rf gpurun_out/r2h/prof_$w; tail -1 gpurun_out/r2h/prof_$w.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value'],1), 'steps/s', {k:v for k,v in d['prefilter'].items() if k in ('k_stages','prefix_exits','candidates')})")
done
