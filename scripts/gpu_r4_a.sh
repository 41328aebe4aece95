#!/bin/bash
# Round 4, FIRST call when a GPU is available (none was during rounds 3 and 4): is HEAD driver-green, and which of the opt-in
# forms built blind this round pay?  (1) the whole GPU suite, every failure listed (not -x); (2) smoke; (3) the driver's exact
# bench command; (4) the opt-in matrix at the headline size, interleaved repetitions; (5) rocprofv3 kernel statistics of the
# default step and of all opt-ins together.  Outputs under gpurun_out/r4a; copy what is to be judged to profiles/r04_*.
O=gpurun_out/r4a; mkdir -p $O
timeout 1900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; tail -c 1200 $O/bench_driver.json
# (the driver's command selects the execution forms on the box, --forms auto: its table is in the line; the same line with the library's defaults)
python -c "
import json; d=json.load(open('$O/bench_driver.json')); fs=d.get('form_selection',{})
print('forms chosen:', fs.get('forms'), '| default', fs.get('default_us'), 'us -> chosen', fs.get('chosen_us'), 'us | crashed', fs.get('crashed'))
for r in fs.get('table',[]): print('  %-40s %8s us identical=%s errors=%s' % (r.get('name'), r.get('us_per_step'), r.get('identical'), r.get('errors')))" 2>/dev/null
timeout 1200 python3 bench.py --gpus 1 --steps 200 --warmup 5 --forms default > $O/bench_default_forms.json 2> $O/bench_default_forms.err; echo "bench (default forms) rc=$?"; python -c "import json; d=json.load(open('$O/bench_default_forms.json')); print(round(d['value']), 'steps/s', d['id_match'].get('frames_identical'))" 2>/dev/null
one() {   # name, env assignments...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --forms default --no-cpu --no-export --steps 200 > $O/opt_${name}_$rep.json 2> $O/opt_${name}_$rep.err
  echo "$name rep $rep: $(python -c "
import json; d=json.load(open('$O/opt_${name}_$rep.json')); s=d['segments_us']
print(round(d['value']), 'steps/s;', ' '.join('%s %.1f' % (k, s[k]) for k in ('prep','k_filter','k_edges1','k_lap','apply1','stage2','merge','dedupe')), '| exits', d['prefilter'].get('prefix_exits'), 'stages', d['prefilter'].get('k_stages'))" 2>/dev/null)"
}
for rep in 1 2; do
  one default BTS_NONE=1
  one first16_5 BTS_I8_FIRST16=5
  one lap_row BTS_LAP_FORM=row
  one sorted BTS_EDGES_SORTED=1
  one ahead BTS_LISTS_AHEAD=1
  one cvt BTS_CVT_OVERLAP=1
  one ahead_cvt BTS_LISTS_AHEAD=1 BTS_CVT_OVERLAP=1
  one predict_first BTS_PREDICT_FIRST=1
  one ahead_predict_front BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1
  one merge_early BTS_MERGE_EARLY=1
  one renorm_fused BTS_RENORM_FUSED=1
  one filter_guess BTS_FILTER_GUESS=1
  one all_r4 BTS_EDGES_SORTED=1 BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1 BTS_CVT_OVERLAP=1 BTS_MERGE_EARLY=1 BTS_RENORM_FUSED=1 BTS_FILTER_GUESS=1
  one all BTS_EDGES_SORTED=1 BTS_LISTS_AHEAD=1 BTS_CVT_OVERLAP=1 BTS_PREDICT_FIRST=1
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -- python bench.py --forms default --steps 100 --no-cpu --no-export > $O/prof_default.log 2>&1
python scripts/prof_top.py $O/prof_default > $O/target_kernel_stats.csv 2>/dev/null; head -16 $O/target_kernel_stats.csv
BTS_EDGES_SORTED=1 BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1 BTS_CVT_OVERLAP=1 BTS_MERGE_EARLY=1 BTS_RENORM_FUSED=1 BTS_FILTER_GUESS=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_all -- python bench.py --forms default --steps 100 --no-cpu --no-export > $O/prof_all.log 2>&1
python scripts/prof_top.py $O/prof_all > $O/target_all_kernel_stats.csv 2>/dev/null; head -16 $O/target_all_kernel_stats.csv
rm -rf $O/prof_default $O/prof_all
