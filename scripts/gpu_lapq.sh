# quick look at k_lap's share: bench (no CPU leg) at the four sizes
for w in ${@:-c2 c3 target c5}; do timeout 300 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$w\", round(d[\"value\"]), d[\"segments_us\"][\"k_lap\"], d[\"segments_us\"][\"step\"])"; done
