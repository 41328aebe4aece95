R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in ${FILLS:-32 64 128 256}; do for w in ${WL:-c2 c3}; do
  BTS_FILL=$g timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats/g -o g -- python $R/bench.py --workload $w --steps 30 --no-cpu > /dev/null 2>&1
  (cd $R && echo "== fill $g $w" && python scripts/prof_top.py gpurun_out/stats/g | grep -E "k_filter|tk_edges<1"; rm -rf gpurun_out/stats/g)
done; done
