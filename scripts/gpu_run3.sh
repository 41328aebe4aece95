mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/t.log
tail -3 gpurun_out/t.log
for w in c2 c3 target; do
  timeout 600 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_$w.json'))
r=d['roofline']
print('$w', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1e3,1), 'us/step | roofline', r['kernel'], r['bound'], round(r['achieved'],2), r['unit'], 'frac', round(r['frac'],4), 'avg_us', round(r['avg_us'],1))
print('   ', {k:round(v,1) for k,v in d.get('segments_us',{}).items()})
PY
done
