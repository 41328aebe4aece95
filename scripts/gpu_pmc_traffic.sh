# HBM traffic per launch of every kernel of a step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only),
# appended to gpurun_out/final/pmc_traffic.txt in the format scripts/pmc_to_json.py reads.  usage: gpu_pmc_traffic.sh c3 c5
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for w in ${@:-c2 target}; do
    timeout 900 rocprofv3 --pmc $c -d $R/gpurun_out/final/pmc_${c}_$w -o p -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/final/pmc_${c}_$w.log 2>&1
    (cd $R && python - <<PY >> gpurun_out/final/pmc_traffic.txt
import glob, sqlite3
dbs = glob.glob("gpurun_out/final/pmc_${c}_$w/**/*.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
print("# $w  $c (KiB per launch, raw counter; see MI355X_MICROARCH.md for the gfx950 x2 correction of wide streaming reads)")
for r in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name='$c' group by kernel_name order by avg(value) desc limit 12"):
    print("%-70s %12.1f  (%d launches)" % (r[0].replace('(anonymous namespace)::','')[:70], r[1], r[2]))
PY
    rm -rf gpurun_out/final/pmc_${c}_$w)
  done
done
cd $R; cat gpurun_out/final/pmc_traffic.txt
