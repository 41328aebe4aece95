#!/bin/bash
mkdir -p gpurun_out/r2i
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_round2.py -q -m gpu -p no:cacheprovider > gpurun_out/r2i/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r2i/tests.log | tail -8
cd /tmp && export TMPDIR=/tmp
for w in target c3 c5 crowd; do for tiles in 0; do
  if [ $tiles = 1 ]; then export BTS_EDGES_TILES=1; else unset BTS_EDGES_TILES; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2i/prof -o p -- python $R/bench.py --workload $w --steps 100 --no-cpu --no-export > $R/gpurun_out/r2i/prof_${w}_$tiles.log 2> $R/gpurun_out/r2i/prof_${w}_$tiles.err
  (cd $R && python scripts/prof_top.py gpurun_out/r2i/prof | grep -E "tk_edges|tk_front" | head -3 | sed "s/^/$w tiles=$tiles /"; rm -rf gpurun_out/r2i/prof; tail -1 gpurun_out/r2i/prof_${w}_$tiles.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value'],1), 'steps/s')")
done; done
