mkdir -p gpurun_out
for w in ${WL:-c2 target}; do
  timeout 600 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu > gpurun_out/bd_$w.json 2> gpurun_out/bd_$w.err || tail -5 gpurun_out/bd_$w.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bd_$w.json'))
print('$w', round(d['value'],1), 'steps/s', round(d['ms_per_step']*1e3,1), 'us/step', d['roofline']['kernel'], round(d['roofline']['avg_us'],1))
print('   ', d['segments_us'])
PY
done
