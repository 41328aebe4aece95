#!/bin/bash
# Round 4, second call (after gpu_r4_a.sh): the measurements.  Bench lines of every workload, the cost of the solver forms on the
# crowd scenes, k_lap's and k_filter_i8's own phase accounting (ab/trace.so, ab/ftrace.so = scripts/build_variant.sh ...), the
# device tie survey and tie_mode's cost, counter passes (HBM traffic per launch).
O=gpurun_out/r4b; mkdir -p $O
for w in target c2 c3 c5 crowd hard-reid dup-box@c2; do
  extra="--no-cpu"; [ $w = target ] && extra=""
  timeout 900 python bench.py --workload $w $extra > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_$w.json')); print(round(d['value']), 'steps/s', d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'us', d.get('id_match',{}).get('frames_identical'))" 2>/dev/null)"
done
for w in crowd crowd-hard-reid c3 target; do for rep in 1 2; do for form in col row; do
  BTS_LAP_FORM=$form timeout 600 python bench.py --forms default --workload $w --no-cpu --no-export --steps 200 > $O/lap_${w}_${form}_$rep.json 2>/dev/null
  echo "$w BTS_LAP_FORM=$form rep $rep: $(python -c "import json; d=json.load(open('$O/lap_${w}_${form}_$rep.json')); print(round(d['value']), 'steps/s; k_lap stage 1', d['lap']['stage1_us'], 'us')" 2>/dev/null)"
done; done; done
if [ -f ab/trace.so ]; then BTS_LIB_PATH=$PWD/ab/trace.so timeout 600 python scripts/dbg_laptrace.py crowd c3 target > $O/lap_trace.txt 2>&1; cat $O/lap_trace.txt; fi
if [ -f ab/ftrace.so ]; then
  BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target > $O/filter_trace.txt 2>&1
  BTS_PREDICT_FIRST=1 BTS_LIB_PATH=$PWD/ab/ftrace.so timeout 300 python scripts/dbg_filtertrace.py target >> $O/filter_trace.txt 2>&1
  cat $O/filter_trace.txt
fi
timeout 900 python scripts/tie_survey.py 2>&1 | grep -v amdgpu.ids > $O/tie_survey.txt; tail -30 $O/tie_survey.txt
timeout 600 python - > $O/tie_mode_cost.txt 2>&1 <<'PY'
import importlib, time, numpy as np, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
for wl in ("c2", "dup-box@c2", "target"):
    K, M, D = pkg.synth.workload_dims(wl)
    for mode in (0, 1):
        st = pkg.synth.make_workload(wl, seed=5)
        trk = pkg.DeviceTracker(max_tracks=K + max(256, K // 8), max_dets=max(K, M), feat_dim=D, tie_mode=mode)
        n = 120 if K <= 1024 else 40
        fr = [st.next() for _ in range(n)]
        dev = [(torch.from_numpy(f.boxes).cuda(), torch.from_numpy(f.scores).cuda(), torch.from_numpy(f.feats).cuda()) for f in fr]
        for f in dev[:n // 6]: trk.step(*f)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in dev[n // 6:]: trk.step(*f)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tied = sum(int(trk.debug_lap(s)[40]) for s in (1, 2, 3))
        print("%-12s tie_mode %d: %.1f us/step (last frame: %d tied stages, errors %d)" % (wl, mode, 1e6 * dt / (n - n // 6), tied, trk.counts()["errors"]))
        trk.close()
PY
cat $O/tie_mode_cost.txt
rm -f gpurun_out/final/pmc_traffic.txt; bash scripts/gpu_pmc_traffic.sh target c2 > $O/pmc.log 2>&1; cp gpurun_out/final/pmc_traffic.txt $O/pmc_traffic.txt 2>/dev/null
python scripts/pmc_to_json.py $O/pmc_traffic.txt && python scripts/kernel_hashes.py --stamp profiles/pmc_traffic.json bot-sort-onnx-tensorrt_amd/libbotsort_hip.so "$(cat $O/../commit.txt 2>/dev/null || echo HEAD)"; cp profiles/pmc_traffic.json $O/pmc_traffic.json; head -c 600 $O/pmc_traffic.json
for w in target crowd; do for h in 15 31 63 127; do
  BTS_LAP_HELPERS=$h timeout 600 python bench.py --forms default --workload $w --no-cpu --no-export --steps 200 > $O/helpers_${w}_$h.json 2>/dev/null
  echo "$w BTS_LAP_HELPERS=$h: $(python -c "import json; d=json.load(open('$O/helpers_${w}_$h.json')); print(round(d['value']), 'steps/s; k_lap stage 1', d['lap']['stage1_us'], 'us')" 2>/dev/null)"
done; done
# config 5's arithmetic mode: what the float32 state / binary16 costs change in time and in HBM traffic at 16384 x 4096 x 384
timeout 600 python - > $O/c5_mode.txt 2>&1 <<'PY'
import importlib, time, torch
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
K, M, D = 16384, 4096, 384
for kd, cd in ((0, 0), (1, 2)):
    st = pkg.synth.SyntheticStream(K, M, D, seed=1234)
    trk = pkg.DeviceTracker(max_tracks=K + K // 8, max_dets=K, feat_dim=D, kalman_dtype=kd, cost_dtype=cd)
    dev = [(lambda f: (torch.from_numpy(f.boxes).cuda(), torch.from_numpy(f.scores).cuda(), torch.from_numpy(f.feats).cuda()))(st.next()) for _ in range(60)]
    for f in dev[:10]: trk.step(*f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in dev[10:]: trk.step(*f)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("kalman_dtype %d cost_dtype %d: %.1f us/step, errors %d" % (kd, cd, 1e6 * dt / 50, trk.counts()["errors"]))
    trk.close()
PY
cat $O/c5_mode.txt
