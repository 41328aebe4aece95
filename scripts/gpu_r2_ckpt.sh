#!/bin/bash
# round-2 checkpoint: GPU suite + smoke, rocprofv3 kernel statistics per workload, FETCH/WRITE counter passes, bench lines.
# usage: gpu_r2_ckpt.sh <tag> [tests|notests]
T=${1:-r2k}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
if [ "${2:-tests}" = tests ]; then
  timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"
  grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -12
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
(cd /tmp && export TMPDIR=/tmp && for w in target c2 c3 c5 crowd hard-reid; do
  g=""; [ $w = c3 ] && g="--motion-gate"
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o $w -- python $R/bench.py --workload $w $g --steps 100 --no-cpu --no-export > $R/$O/prof_$w.log 2>&1
  (cd $R && python scripts/prof_top.py $O/prof_$w > $O/${w}_kernel_stats.csv; echo "== $w"; head -14 $O/${w}_kernel_stats.csv; rm -rf $O/prof_$w)
done
for c in FETCH_SIZE WRITE_SIZE; do
  for w in target c2; do
    timeout 900 rocprofv3 --pmc $c -d $R/$O/pmc_${c}_$w -o p -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --no-export > $R/$O/pmc_${c}_$w.log 2>&1
    (cd $R && python - <<PY >> $O/pmc_traffic.txt
import glob, sqlite3
dbs = glob.glob("$O/pmc_${c}_$w/**/*.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
print("# $w  $c (KiB per launch, raw counter; see MI355X_MICROARCH.md for the gfx950 x2 correction of wide streaming reads)")
for r in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name='$c' group by kernel_name order by avg(value) desc limit 14"):
    print("%-70s %12.1f  (%d launches)" % (r[0].replace('(anonymous namespace)::','')[:70], r[1], r[2]))
PY
    rm -rf $O/pmc_${c}_$w)
  done
done)
cat $O/pmc_traffic.txt | head -40
timeout 1200 python bench.py > $O/bench_target.json 2> $O/bench_target.err; echo "bench rc=$?"
for w in c2 c3 c5 crowd hard-reid; do
  g=""; [ $w = c3 ] && g="--motion-gate"
  timeout 900 python bench.py --workload $w $g --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 900 python bench.py --gpus 2 --oversubscribe --no-cpu > $O/bench_target_2ranks_1gpu.json 2> $O/bench_2r.err; echo "2-rank rc=$?"
python - <<PY
import json
for w in ("target","c2","c3","c5","crowd","hard-reid","target_2ranks_1gpu"):
    try:
        d=json.loads(open('$O/bench_%s.json'%w).read().strip().splitlines()[-1])
        print(w, round(d['value'],1),'steps/s', d['roofline']['kernel'][:30], round(d['roofline']['avg_us'],1),'us', 'frac', round(d['roofline']['frac'],3), 'step as-built', round(d['roofline']['step']['frac_as_built'],3), 'survey', round(d['roofline']['step']['frac_survey'],3), 'idm', (d.get('id_match') or {}).get('frames_identical'))
    except Exception as e: print(w, 'ERR', e)
PY
