# per-dispatch durations of k_lap (stage 1 vs stages 2+3 alternate) and inter-kernel gaps from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT; W=${1:-c2}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python $R/bench.py --workload $W --steps 128 --no-cpu $EXTRA > /tmp/lt.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/lt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//2:]
lap = [r for r in rows if 'k_lap' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in lap]
print('k_lap even/odd avg us:', sum(d[0::2])/len(d[0::2])/1e3, sum(d[1::2])/len(d[1::2])/1e3)
g = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g[b['Kernel_Name'][:40]].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for k, v in g.items():
    print('gap before %-42s %.2f us (n=%d)' % (k, sum(v)/len(v)/1e3, len(v)))
PY
