# PMC passes over the kernels matching $KERN (sqlite LIKE pattern) at workload $WL (one counter group per pass)
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d $R/gpurun_out/pmc/p$i -o p -- python $R/bench.py --workload ${WL:-target} --steps 10 --warmup 5 --no-cpu > $R/gpurun_out/pmc/p$i.log 2>&1
  (cd $R && python - <<PY
import glob, sqlite3
dbs = glob.glob("gpurun_out/pmc/p$i/**/*.db", recursive=True)
if not dbs: print("no db for pass $i"); raise SystemExit
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '${KERN:-%tk_edges%}' group by kernel_name, counter_name").fetchall()
for r in rows: print(r[0][-60:], r[1], round(r[2], 1), r[3])
PY
  rm -rf $R/gpurun_out/pmc/p$i)
done
