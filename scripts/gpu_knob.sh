mkdir -p gpurun_out/stats
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in ${KNOBS:-0 1 2 3}; do
  BTS_DEBUG_KNOB=$k timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats/k$k -o k -- python $R/bench.py --workload ${WL:-target} --steps 30 --no-cpu > $R/gpurun_out/stats/k$k.log 2>&1
  (cd $R && echo "== knob $k" && python scripts/prof_top.py gpurun_out/stats/k$k | grep -E "${KPAT:-tk_edges}"; rm -rf gpurun_out/stats/k$k)
done
