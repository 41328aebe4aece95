#!/bin/bash
mkdir -p gpurun_out/r2j
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in target c2; do
  BTS_EDGES_TILES=1 timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r2j/prof -o p -- python $R/bench.py --workload $w --steps 60 --no-cpu --no-export > $R/gpurun_out/r2j/$w.log 2> $R/gpurun_out/r2j/$w.err
  (cd $R && echo "== $w" && python scripts/prof_timeline.py gpurun_out/r2j/prof; rm -rf gpurun_out/r2j/prof)
done
