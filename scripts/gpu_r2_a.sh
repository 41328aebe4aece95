#!/bin/bash
# round-2 checkpoint a: new parity tests, tie survey, headline bench line, spawn path
mkdir -p gpurun_out/r2a
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r2a/tests_round2.log 2>&1; echo "round2 tests rc=$?" 
tail -15 gpurun_out/r2a/tests_round2.log
timeout 300 python scripts/tie_survey.py > gpurun_out/r2a/tie_survey.txt 2>&1; echo "tie rc=$?"
cat gpurun_out/r2a/tie_survey.txt
timeout 900 python bench.py > gpurun_out/r2a/bench_target.json 2> gpurun_out/r2a/bench_target.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r2a/bench_target.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2a/bench_target.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['config']['workload']); print(d['roofline']['step']); print({k:d['roofline'][k] for k in ('kernel','frac','avg_us')})
print(d.get('export_inclusive')); print(d.get('cpu_baseline')); print(d.get('id_match')); print(d['segments_us']); print(d['lap'])
PY
timeout 600 python bench.py --gpus 2 --oversubscribe --workload c2 --steps 100 > gpurun_out/r2a/bench_c2_spawn2.json 2> gpurun_out/r2a/bench_c2_spawn2.err; echo "spawn rc=$?"
tail -c 600 gpurun_out/r2a/bench_c2_spawn2.err; python -c "
import json; d=json.load(open('gpurun_out/r2a/bench_c2_spawn2.json')); print(d['n_gpus'], d['value'], d['config']['spawn'], d.get('id_match'))"
