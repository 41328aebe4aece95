mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/t.log
tail -3 gpurun_out/t.log
for w in c2 c3 target; do
 for g in "" "--graph"; do
  timeout 600 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu $g > gpurun_out/bench_$w$g.json 2> gpurun_out/bench_$w$g.err || tail -5 gpurun_out/bench_$w$g.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/bench_$w$g.json'))
    r=d['roofline']
    print('$w $g', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1e3,1), 'us/step | roofline', r['kernel'], round(r['achieved'],2), r['unit'], 'frac', round(r['frac'],4), 'avg_us', round(r['avg_us'],1))
except Exception as e:
    print('$w $g failed', e)
PY
 done
done
