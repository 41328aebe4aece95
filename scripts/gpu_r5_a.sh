#!/bin/bash
# Round 5, first GPU contact (no GPU was granted during rounds 3, 4 and -- so far -- 5).  The measurement comes first: smoke, the
# headline line with the library's DEFAULT forms (no selection child: nothing never-run is stepped before this number exists), then
# the whole -m gpu suite with per-test durations (the driver's limit for it is 1200 s), then the driver's exact bench command (default
# forms timed first, selection second, both figures in the line), then rocprofv3 kernel statistics of the default step.
O=gpurun_out/r5a; mkdir -p $O
rocminfo | grep -m1 gfx
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 200 --warmup 5 --forms default > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']), 'steps/s', d['ms_per_step'], 'ms/step; id_match', d.get('id_match',{}).get('frames_identical'), '/', d.get('id_match',{}).get('frames'), '; roofline', d['roofline']['kernel'], d['roofline'].get('avg_us'), 'us frac', d['roofline'].get('frac')); print(d['segments_us'])" 2>&1 | tail -3
tail -3 $O/bench_default.err
T0=$(date +%s)
timeout 2100 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=60 > $O/tests.log 2>&1; echo "tests rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -40
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench (driver's command) rc=$?"
python -c "
import json; d=json.load(open('$O/bench_driver.json')); fs=d.get('form_selection',{})
print(round(d['value']), 'steps/s =', d.get('value_is'), '| default', d.get('value_default_forms'), '| selected', d.get('value_selected_forms'))
print('forms chosen:', fs.get('forms'), '| default', fs.get('default_us_both'), 'us, noise', fs.get('noise_pct'), '% -> chosen', fs.get('chosen_us'), 'us | crashed', fs.get('crashed'))
for r in fs.get('table',[]): print('  %-40s %8s us identical=%s errors=%s %s' % (r.get('name'), r.get('us_per_step'), r.get('identical'), r.get('errors'), r.get('verdict')))" 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -- python bench.py --forms default --steps 100 --no-cpu --no-export > $O/prof_default.log 2>&1
python scripts/prof_top.py $O/prof_default > $O/target_kernel_stats.csv 2>/dev/null; head -16 $O/target_kernel_stats.csv
rm -rf $O/prof_default
