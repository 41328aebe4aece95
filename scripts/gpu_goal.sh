R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in ${GOALS:-768 1536 3072 6144}; do for w in ${WL:-c3 target}; do
  BTS_TILE_GOAL=$g timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats/g -o g -- python $R/bench.py --workload $w --steps 30 --no-cpu > /dev/null 2>&1
  (cd $R && echo "== goal $g $w" && python scripts/prof_top.py gpurun_out/stats/g | grep -E "tk_edges<1"; rm -rf gpurun_out/stats/g)
done; done
