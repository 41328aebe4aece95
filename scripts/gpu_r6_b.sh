#!/bin/bash
# Round 6, second call: what BTS_FILTER_AHEAD is worth and what it costs the step's own kernels.  (1) interleaved A/B at the headline
# size, c3 and c2: default forms / filter-ahead alone / with merge-early, fused renormalisation, the SELF solver / with 384 and 512
# side workgroups -- per-segment timings of the STEP's stream in every line (contention shows as k_edges1 / k_mid2 growing);
# (2) the digest soak of the form (2 000 frames, with and without the residency promise) that has to pass before tune.py may offer it
# (BTS_TUNE_ALLOW_CVT=1); (3) rocprofv3 kernel statistics of the form (the side passes appear as k_filter_i8 launches on a second
# queue); (4) round 5's matrix of the other forms, the pre-filter's phase accounting (against DESIGN 7.1's prediction) and the
# counter passes: scripts/gpu_r5_b.sh.
O=gpurun_out/r6b; mkdir -p $O
one() {   # workload, name, env assignments...
  local wl=$1 name=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $wl --forms default --no-cpu --no-export --steps 200 > $O/${wl}_${name}_$rep.json 2> $O/${wl}_${name}_$rep.err
  echo "$wl $name rep $rep: $(python -c "
import json; d=json.load(open('$O/${wl}_${name}_$rep.json')); s=d['segments_us']
print(round(d['value']), 'steps/s;', ' '.join('%s %.1f' % (k, s[k]) for k in ('prep','k_filter','k_edges1','k_lap','apply1','stage2','merge','dedupe') if k in s), '| launch:', d['config'].get('launch'))" 2>/dev/null)"
}
for rep in 1 2 3; do
  for wl in target c3 c2; do
    one $wl default BTS_NONE=1
    one $wl ahead BTS_FILTER_AHEAD=1
    one $wl ahead_all BTS_FILTER_AHEAD=1 BTS_MERGE_EARLY=1 BTS_RENORM_FUSED=1 BTS_LAP_SELF=1
    one $wl ahead_smooth BTS_FILTER_AHEAD=1 BTS_SMOOTH_DEFERRED=1
    one $wl ahead_sorted_smooth BTS_FILTER_AHEAD=1 BTS_SMOOTH_DEFERRED=1 BTS_EDGES_SORTED=1
    one $wl throughput_all BTS_FILTER_AHEAD=1 BTS_SMOOTH_DEFERRED=1 BTS_EDGES_SORTED=1 BTS_MERGE_EARLY=1 BTS_RENORM_FUSED=1 BTS_LAP_SELF=1
    one $wl ahead_pw384 BTS_FILTER_AHEAD=1 BTS_FILTER_AHEAD_PW=384
    one $wl ahead_pw512 BTS_FILTER_AHEAD=1 BTS_FILTER_AHEAD_PW=512
    one $wl lists_front BTS_LISTS_AHEAD=1 BTS_PREDICT_FRONT=1
  done
done
timeout 1500 python scripts/soak_cvt_overlap.py 2000 BTS_FILTER_AHEAD=1 > $O/soak_filter_ahead.txt 2>&1; tail -14 $O/soak_filter_ahead.txt
timeout 1500 python scripts/soak_cvt_overlap.py 1000 BTS_FILTER_AHEAD=1,BTS_SMOOTH_DEFERRED=1,BTS_EDGES_SORTED=1,BTS_MERGE_EARLY=1 > $O/soak_throughput_all.txt 2>&1; tail -14 $O/soak_throughput_all.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BTS_FILTER_AHEAD=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_ahead -- python bench.py --forms default --steps 100 --no-cpu --no-export > $O/prof_ahead.log 2>&1
python scripts/prof_top.py $O/prof_ahead > $O/target_ahead_kernel_stats.csv 2>/dev/null; head -20 $O/target_ahead_kernel_stats.csv
rm -rf $O/prof_ahead
bash scripts/gpu_r5_b.sh
