# Final measurement set of a round: bench JSON (with CPU baseline), rocprof kernel stats, PMC traffic.
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
timeout 900 python bench.py --workload target --steps 100 --no-cpu > gpurun_out/final/bench_target.json 2> gpurun_out/final/bench_target.err
timeout 900 python bench.py --workload c3 --steps 100 --no-cpu > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err
timeout 900 python bench.py --workload c5 --steps 30 --no-cpu > gpurun_out/final/bench_c5.json 2> gpurun_out/final/bench_c5.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_c2 -o c2 -- python $R/bench.py --no-cpu > $R/gpurun_out/final/prof_c2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_target -o tgt -- python $R/bench.py --workload target --steps 50 --no-cpu > $R/gpurun_out/final/prof_target.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/final/pmc_fetch -o f -- python $R/bench.py --workload target --steps 10 --warmup 5 --no-cpu > $R/gpurun_out/final/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/final/pmc_write -o w -- python $R/bench.py --workload target --steps 10 --warmup 5 --no-cpu > $R/gpurun_out/final/pmc_write.log 2>&1
cd $R
python scripts/prof_top.py gpurun_out/final/prof_c2 > gpurun_out/final/c2_kernel_stats.csv
python scripts/prof_top.py gpurun_out/final/prof_target > gpurun_out/final/target_kernel_stats.csv
python - <<'PY'
import glob, sqlite3
for tag in ("pmc_fetch", "pmc_write"):
    dbs = glob.glob("gpurun_out/final/%s/**/*.db" % tag, recursive=True)
    if not dbs:
        print(tag, "no db"); continue
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    print(tag, [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()][:10])
    for t in tabs:
        if t in ("counters_collection", "pmc_events", "counters"):
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            print(t, cols)
            for r in cur.execute("select * from %s limit 3" % t): print(r)
PY
head -8 gpurun_out/final/c2_kernel_stats.csv; head -8 gpurun_out/final/target_kernel_stats.csv
for f in c2 c3 target c5; do python -c "
import json
d=json.load(open('gpurun_out/final/bench_$f.json'))
print('$f', round(d['value'],1), 'steps/s', d['roofline']['kernel'], round(d['roofline']['frac'],4), d.get('cpu_baseline',{}).get('value'))
"; done
