#!/bin/bash
# Round 5, first GPU contact.  The measurement comes first: smoke, the headline line with the library's default forms, then the
# whole -m gpu suite with per-test durations (the driver's limit for it is 1200 s), then the driver's exact bench command.
O=gpurun_out/r5a; mkdir -p $O
rocminfo | grep -m1 gfx; 
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 200 --warmup 5 --forms default > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
T0=$(date +%s)
timeout 2100 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=60 > $O/tests.log 2>&1; echo "tests rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -40
