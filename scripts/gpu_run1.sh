mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/t3.log
timeout 600 python bench.py --breakdown > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 600 python bench.py --workload target --steps 30 --warmup 5 --breakdown --no-cpu > gpurun_out/bench_target.json 2> gpurun_out/bench_target.err
timeout 600 python bench.py --workload c3 --steps 50 --warmup 5 --breakdown --no-cpu > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/t3.log; cat gpurun_out/bench_c2.json gpurun_out/bench_c2.err gpurun_out/bench_target.json gpurun_out/bench_target.err gpurun_out/bench_c3.json gpurun_out/bench_c3.err | cut -c1-1500
ls -R gpurun_out/prof_c2 | head -20
