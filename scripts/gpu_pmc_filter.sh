# PMC passes over the pre-filter kernel at the target workload (one counter group per pass)
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d $R/gpurun_out/pmc/p$i -o p -- python $R/bench.py --workload ${WL:-target} --steps 10 --warmup 5 --no-cpu > $R/gpurun_out/pmc/p$i.log 2>&1
  (cd $R && python - <<PY
import glob, sqlite3
dbs = glob.glob("gpurun_out/pmc/p$i/**/*.db", recursive=True)
if not dbs: print("no db for pass $i"); raise SystemExit
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_filter%' group by kernel_name, counter_name").fetchall()
for r in rows: print(r[0][:40], r[1], round(r[2], 1), r[3])
PY
  rm -rf $R/gpurun_out/pmc/p$i)
done
