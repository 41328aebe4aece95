#!/bin/bash
mkdir -p gpurun_out/clk
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30
( BTS_BENCH_IGNORE_ERRORS=1 python bench.py --workload target --no-cpu --no-export --steps 4000 --roofline-steps 8 > gpurun_out/clk/b1.json 2> gpurun_out/clk/b1.err ) &
sleep 45
for i in 1 2 3 4; do rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -4; sleep 2; done
wait
python -c "
import json; d=json.load(open('gpurun_out/clk/b1.json')); print('default perf level:', round(d['value'],1), 'steps/s', d['segments_us']['k_filter'])"
echo "--- set perf level high"
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel --showclocks 2>&1 | grep -E "Performance|sclk" | head -4
BTS_BENCH_IGNORE_ERRORS=1 python bench.py --workload target --no-cpu --no-export --steps 200 > gpurun_out/clk/b2.json 2> gpurun_out/clk/b2.err
python -c "
import json; d=json.load(open('gpurun_out/clk/b2.json')); print('perf level high:', round(d['value'],1), 'steps/s', d['segments_us'])"
rocm-smi --setperflevel auto 2>&1 | tail -1
