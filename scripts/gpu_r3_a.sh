#!/bin/bash
# Round 3, first call when a GPU is available: is HEAD driver-green?  Whole GPU suite (all failures listed, not -x), smoke,
# the driver's exact bench command, the device tie survey.  Outputs under gpurun_out/r3a; copy what is to be judged to profiles/.
O=gpurun_out/r3a; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -15
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; tail -c 1500 $O/bench_driver.json
timeout 900 python scripts/tie_survey.py 2>&1 | grep -v amdgpu.ids > $O/tie_survey.txt; tail -30 $O/tie_survey.txt
