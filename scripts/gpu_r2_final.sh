#!/bin/bash
# end-of-round batch: tie survey (kept this time), the whole GPU suite + smoke, one default bench line
O=gpurun_out/r2z; mkdir -p $O
python scripts/tie_survey.py 2>&1 | grep -v amdgpu.ids > $O/tie_survey.txt; cat $O/tie_survey.txt
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > $O/bench_target.json 2> $O/bench_target.err; echo "bench rc=$?"; tail -c 300 $O/bench_target.json
