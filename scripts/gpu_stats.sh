# rocprofv3 kernel stats for the workloads in $WL (default: c2 target)
mkdir -p gpurun_out/stats
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in ${WL:-c2 target}; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats/prof_$w -o $w -- python $R/bench.py --workload $w --steps 100 --no-cpu > $R/gpurun_out/stats/prof_$w.log 2>&1
  (cd $R && python scripts/prof_top.py gpurun_out/stats/prof_$w > gpurun_out/stats/${w}_kernel_stats.csv; echo "== $w"; cat gpurun_out/stats/${w}_kernel_stats.csv; rm -rf gpurun_out/stats/prof_$w)
done
