#!/bin/bash
# Round 3, second call (after gpu_r3_a.sh is green): the measurements.  Bench lines of every workload incl. the new ones,
# rocprofv3 kernel stats at the headline size, A/B of the merged tall-noaccum kernels against the round-2 library (ab/head.so),
# cost of tie_mode 1 on a tie-free stream and on the dup-box scene.
O=gpurun_out/r3b; mkdir -p $O
for w in target c2 c3 c5 crowd hard-reid dup-box@c2; do
  extra="--no-cpu"; [ $w = target ] && extra=""
  timeout 900 python bench.py --workload $w $extra > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_$w.json')); print(round(d['value']), 'steps/s', d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'us', 'facade' in d)" 2>/dev/null)"
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_target -- python bench.py --steps 100 --no-cpu --no-export > $O/prof_target.log 2>&1
python scripts/prof_top.py $O/prof_target > $O/target_kernel_stats.csv 2>/dev/null; head -20 $O/target_kernel_stats.csv
# A/B against the library of the last commit that ran on hardware in round 2 (ab/r2_8cbc337.so, built from a worktree of 8cbc337;
# ab/head.so = scripts/build_variant.sh head): the three changed kernels of the default step
[ -f ab/r2_8cbc337.so ] && [ -f ab/head.so ] && for w in target c3; do echo "== A/B $w"; VARIANTS="r2_8cbc337 head" WL=$w KERN="tk_front|tk_edges_tall|tk_mid1|k_filter" bash scripts/gpu_ab2.sh 2>&1 | tail -24; done
# k_lap: column-form / whole-LDS component solvers (round 3, default) against the round-2 row forms
# (ab/lap_rowform.so = scripts/build_variant.sh lap_rowform EXTRA=-DBTS_LAP_TALL_ROWFORM), on the crowd scenes and the headline size
# (lap_rowform shares round 3's preparation layouts: it isolates the solvers; r2_8cbc337 is the whole round-2 library)
[ -f ab/lap_rowform.so ] && [ -f ab/head.so ] && for w in crowd crowd-hard-reid target; do echo "== A/B k_lap $w"; VARIANTS="$([ -f ab/r2_8cbc337.so ] && echo r2_8cbc337) lap_rowform head" WL=$w KERN="k_lap" bash scripts/gpu_ab2.sh 2>&1 | tail -18; done
# k_lap's own phase timeline (ab/trace.so = scripts/build_variant.sh trace TRACE=1): where the preparation and the solve spend the launch
if [ -f ab/trace.so ]; then BTS_LIB_PATH=$PWD/ab/trace.so timeout 600 python scripts/dbg_laptrace.py crowd c3 target > $O/lap_trace.txt 2>&1; cat $O/lap_trace.txt; fi
# scheduling experiments (same library, env switches read at bts_tracker_create), interleaved, at the headline size and c3
for w in target c3; do for rep in 1 2; do for exp in none BTS_PREDICT_FIRST BTS_CVT_OVERLAP both; do
  unset BTS_PREDICT_FIRST BTS_CVT_OVERLAP
  [ $exp = BTS_PREDICT_FIRST -o $exp = both ] && export BTS_PREDICT_FIRST=1
  [ $exp = BTS_CVT_OVERLAP -o $exp = both ] && export BTS_CVT_OVERLAP=1
  timeout 600 python bench.py --workload $w --no-cpu --no-export --steps 300 > $O/exp_${w}_${exp}_$rep.json 2>/dev/null
  echo "$w $exp rep $rep: $(python -c "import json; d=json.load(open('$O/exp_${w}_${exp}_$rep.json')); s=d['segments_us']; print(round(d['value']), 'steps/s; prep', round(s['prep'],1), 'k_filter', round(s['k_filter'],1), 'us')" 2>/dev/null)"
done; done; done; unset BTS_PREDICT_FIRST BTS_CVT_OVERLAP
# where k_filter_i8's time goes (its own cycle accounting; ab/ftrace.so = scripts/build_variant.sh ftrace FTRACE=1)
if [ -f ab/ftrace.so ]; then
  BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target > $O/filter_trace.txt 2>&1
  BTS_PREDICT_FIRST=1 BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target >> $O/filter_trace.txt 2>&1
  cat $O/filter_trace.txt
fi
timeout 600 python - > $O/tie_mode_cost.txt 2>&1 <<'PY'
import importlib, time, numpy as np, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
for wl, kw in (("c2", {}), ("dup-box@c2", {})):
    K, M, D = pkg.synth.workload_dims(wl)
    for mode in (0, 1):
        st = pkg.synth.make_workload(wl, seed=5)
        trk = pkg.DeviceTracker(max_tracks=K + 256, max_dets=max(K, M), feat_dim=D, tie_mode=mode)
        fr = [st.next() for _ in range(120)]
        dev = [(torch.from_numpy(f.boxes).cuda(), torch.from_numpy(f.scores).cuda(), torch.from_numpy(f.feats).cuda()) for f in fr]
        for f in dev[:20]: trk.step(*f)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in dev[20:]: trk.step(*f)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tied = sum(int(trk.debug_lap(s)[40]) for s in (1, 2, 3))
        print("%-12s tie_mode %d: %.1f us/step (last frame: %d tied stages, errors %d)" % (wl, mode, 1e6 * dt / 100, tied, trk.counts()["errors"]))
        trk.close()
PY
cat $O/tie_mode_cost.txt
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE, separate counter-only passes) -> profiles/pmc_traffic.json for roofline.traffic
rm -f gpurun_out/final/pmc_traffic.txt; bash scripts/gpu_pmc_traffic.sh target c2 > $O/pmc.log 2>&1; cp gpurun_out/final/pmc_traffic.txt $O/pmc_traffic.txt 2>/dev/null
python scripts/pmc_to_json.py $O/pmc_traffic.txt; cp profiles/pmc_traffic.json $O/pmc_traffic.json; head -c 600 $O/pmc_traffic.json
# host-side sweep knobs (no rebuild): helper workgroups of k_lap, persistent workgroups of k_filter_i8
for w in target crowd; do for h in 15 31 63 127; do
  BTS_LAP_HELPERS=$h timeout 600 python bench.py --workload $w --no-cpu --no-export --steps 200 > $O/helpers_${w}_$h.json 2>/dev/null
  echo "$w BTS_LAP_HELPERS=$h: $(python -c "import json; d=json.load(open('$O/helpers_${w}_$h.json')); print(round(d['value']), 'steps/s; k_lap stage 1', d['lap']['stage1_us'], 'us')" 2>/dev/null)"
done; done
for pw in 369 448 512 553; do
  BTS_FILTER_PW=$pw timeout 600 python bench.py --no-cpu --no-export --steps 200 > $O/pw_$pw.json 2>/dev/null
  echo "target BTS_FILTER_PW=$pw: $(python -c "import json; d=json.load(open('$O/pw_$pw.json')); print(round(d['value']), 'steps/s; k_filter', round(d['segments_us']['k_filter'],1), 'us')" 2>/dev/null)"
done
