#!/bin/bash
# Round 5, second call: what the pre-filter's forms are worth.  (1) the form matrix at the headline size with per-segment timings,
# interleaved repetitions (tile shapes of round 5 next to round 4's forms); (2) k_filter_i8's own phase accounting from the trace
# variant (ab/ftrace.so = scripts/build_variant.sh ftrace FTRACE=1): the decomposition of the "fixed 22.5 us" -> copy to
# profiles/r05_kfilter_phases.txt; (3) the 2 000-frame digest soak of BTS_CVT_OVERLAP (with and without the residency promise) that
# has to pass before tune.py may offer it (BTS_TUNE_ALLOW_CVT=1); (4) counter passes stamped with HEAD's kernel hashes.
O=gpurun_out/r5b; mkdir -p $O
one() {   # name, env assignments...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --forms default --no-cpu --no-export --steps 200 > $O/opt_${name}_$rep.json 2> $O/opt_${name}_$rep.err
  echo "$name rep $rep: $(python -c "
import json; d=json.load(open('$O/opt_${name}_$rep.json')); s=d['segments_us']
print(round(d['value']), 'steps/s;', ' '.join('%s %.1f' % (k, s[k]) for k in ('prep','k_filter','k_edges1','k_lap','apply1','stage2','merge','dedupe')), '| tile', d['prefilter'].get('tile'), 'tiles', d['prefilter'].get('tiles'), 'exits', d['prefilter'].get('prefix_exits'), 'stages', d['prefilter'].get('k_stages'))" 2>/dev/null)"
}
for rep in 1 2 3; do
  one default BTS_NONE=1
  one tile160 BTS_FILTER_TILE=160x128
  one tile64 BTS_FILTER_TILE=128x64
  one tile160_guess BTS_FILTER_TILE=160x128 BTS_FILTER_GUESS=1
  one predict_first BTS_PREDICT_FIRST=1
  one ahead_predict_front BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1
  one tile160_ahead_front BTS_FILTER_TILE=160x128 BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1
  one planes BTS_SHADOW_PLANES=1
  one tile160_planes BTS_FILTER_TILE=160x128 BTS_SHADOW_PLANES=1
  one sorted BTS_EDGES_SORTED=1
  one merge_early BTS_MERGE_EARLY=1
  one renorm_fused BTS_RENORM_FUSED=1
  one lap_self BTS_LAP_SELF=1
  one all_r5 BTS_FILTER_TILE=160x128 BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1 BTS_MERGE_EARLY=1 BTS_RENORM_FUSED=1 BTS_FILTER_GUESS=1 BTS_LAP_SELF=1
done
if [ -f ab/ftrace.so ]; then
  BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target > $O/kfilter_phases.txt 2>&1
  BTS_PREDICT_FIRST=1 BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target >> $O/kfilter_phases.txt 2>&1
  BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py c3 >> $O/kfilter_phases.txt 2>&1
  cat $O/kfilter_phases.txt
fi
timeout 1500 python scripts/soak_cvt_overlap.py 2000 > $O/soak_cvt.txt 2>&1; tail -6 $O/soak_cvt.txt
rm -f gpurun_out/final/pmc_traffic.txt; bash scripts/gpu_pmc_traffic.sh target c2 > $O/pmc.log 2>&1; cp gpurun_out/final/pmc_traffic.txt $O/pmc_traffic.txt 2>/dev/null
python scripts/pmc_to_json.py $O/pmc_traffic.txt && python scripts/kernel_hashes.py --stamp profiles/pmc_traffic.json bot-sort-onnx-tensorrt_amd/libbotsort_hip.so "$(cat $O/../commit.txt 2>/dev/null || echo HEAD)"; cp profiles/pmc_traffic.json $O/pmc_traffic.json; head -c 600 $O/pmc_traffic.json
