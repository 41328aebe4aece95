# usage: gpu_prof.sh <workload> <steps> <tag>
W=$1; S=$2; TAG=$3
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps $S --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/prof_top.py gpurun_out/prof_$TAG > gpurun_out/prof_$TAG.csv
cat gpurun_out/prof_$TAG.csv
