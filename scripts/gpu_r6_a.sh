#!/bin/bash
# Round 6, first GPU contact since mid-round 2.  Measurement first: smoke, the headline line with the library's DEFAULT forms (no
# selection child: nothing optional is stepped before this number exists); if either fails, the triage leg (per-frame comparison
# with the oracle + the solver's / pre-filter's counters at tiny, c1, c2) so the lease yields a diagnosis; then the whole -m gpu
# suite WITHOUT -x, with -rf --tb=short and per-test durations; then the driver's exact bench command (on-device form selection with
# the side-stream forms' soak); then rocprofv3 kernel statistics of the default step.
O=gpurun_out/r6a; mkdir -p $O
rocminfo | grep -m1 gfx
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; SMOKE=$?; tail -3 $O/smoke.log; echo "smoke rc=$SMOKE"
timeout 900 python3 bench.py --gpus 1 --steps 200 --warmup 5 --forms default > $O/bench_default.json 2> $O/bench_default.err; BENCH=$?; echo "bench default rc=$BENCH"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']), 'steps/s', d['ms_per_step'], 'ms/step; id_match', d.get('id_match',{}).get('frames_identical'), '/', d.get('id_match',{}).get('frames'), '; roofline', d['roofline']['kernel'], d['roofline'].get('avg_us'), 'us frac', d['roofline'].get('frac')); print(d['segments_us'])" 2>&1 | tail -3
tail -5 $O/bench_default.err
if [ $SMOKE -ne 0 ] || [ $BENCH -ne 0 ]; then
  for w in tiny c1 c2; do timeout 300 python scripts/triage_first_contact.py $w 3 > $O/triage_$w.log 2>&1; echo "triage $w rc=$?"; tail -25 $O/triage_$w.log; done
fi
T0=$(date +%s)
timeout 2100 python -m pytest tests -q -m gpu -p no:cacheprovider -rf --tb=short --durations=40 > $O/tests.log 2>&1; echo "tests rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "passed|failed|^FAILED|^ERROR" $O/tests.log | tail -60
# the driver's exact command: default forms timed first, then the on-device selection (incl. the side-stream forms' soak, round 6),
# the selected forms re-timed on the same frames; both figures, every candidate's verdict and the soak records in the line
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench (driver's command) rc=$?"
python scripts/first_contact_report.py $O/bench_driver.json 2>/dev/null | head -70
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -- python bench.py --forms default --steps 100 --no-cpu --no-export > $O/prof_default.log 2>&1
python scripts/prof_top.py $O/prof_default > $O/target_kernel_stats.csv 2>/dev/null; head -20 $O/target_kernel_stats.csv
rm -rf $O/prof_default
