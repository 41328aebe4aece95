#!/bin/bash
# within-box A/B of library builds: ab/<name>.so, interleaved, rocprofv3 kernel statistics at the headline size
mkdir -p gpurun_out/ab
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in ${VARIANTS:-head prio}; do
  BTS_LIB_PATH=$R/ab/$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab/prof_$v -o p -- python $R/bench.py --workload ${WL:-target} --steps 100 --no-cpu --no-export > $R/gpurun_out/ab/$v.log 2> $R/gpurun_out/ab/$v.err
  (cd $R && python scripts/prof_top.py gpurun_out/ab/prof_$v | grep -E "${KERN:-k_filter}" | head -3 | sed "s/^/$v /"; rm -rf gpurun_out/ab/prof_$v; tail -1 gpurun_out/ab/$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['value'],1), 'steps/s')")
done; done
