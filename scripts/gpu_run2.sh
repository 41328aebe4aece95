mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/t.log
tail -15 gpurun_out/t.log
for w in c2 c3 target; do
  timeout 600 python bench.py --workload $w --steps 60 --warmup 5 --breakdown --no-cpu > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_$w.json'))
print('$w', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v,1) for k,v in d.get('segments_us',{}).items()})
PY
done
