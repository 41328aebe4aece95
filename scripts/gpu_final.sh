# Measurement set of a checkpoint: bench JSONs (c2 with the CPU baseline), rocprof kernel stats, PMC traffic of the
# dominant kernels.  Results land in gpurun_out/final/ ; copy what is to be kept into profiles/.
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
for w in c3 target c5; do
  timeout 900 python bench.py --workload $w --steps 100 --no-cpu > gpurun_out/final/bench_$w.json      # (c3 runs with the motion gate, as BASELINE states it) 2> gpurun_out/final/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
for w in c2 target; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_$w -o $w -- python $R/bench.py --workload $w --steps 100 --no-cpu > $R/gpurun_out/final/prof_$w.log 2>&1
  (cd $R && python scripts/prof_top.py gpurun_out/final/prof_$w > gpurun_out/final/${w}_kernel_stats.csv; rm -rf gpurun_out/final/prof_$w)
done
for c in FETCH_SIZE WRITE_SIZE; do
  for w in c2 target; do
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
cd $R
head -14 gpurun_out/final/c2_kernel_stats.csv; head -14 gpurun_out/final/target_kernel_stats.csv
for f in c2 c3 target c5; do python -c "
import json
d=json.load(open('gpurun_out/final/bench_$f.json'))
print('$f', round(d['value'],1), 'steps/s', d['roofline']['kernel'], 'frac', round(d['roofline']['frac'],4), 'avg_us', round(d['roofline']['avg_us'],1), d.get('cpu_baseline',{}).get('value'))
"; done
cat gpurun_out/final/pmc_traffic.txt | head -60
