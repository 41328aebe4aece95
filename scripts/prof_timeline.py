"""Timeline of one steady-state step from a rocprofv3 kernel trace (rocpd sqlite): start offset, duration, gap to the previous kernel."""
import glob, sqlite3, sys
path = sys.argv[1]
db = sqlite3.connect(glob.glob(path + "/**/*.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
sym = [t for t in tabs if "kernel_symbol" in t]
q = "select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (kt, sym[0]) if sym and "kernel_id" in cols else None
rows = list(cur.execute(q)) if q else []
if not rows:
    print("tables", tabs, "cols", cols); sys.exit(0)
# find the last occurrence of tk_front and print from there to the next one
idx = [k for k, r in enumerate(rows) if "tk_front" in r[2]]
a, b = idx[-3], idx[-2]
t0 = rows[a][0]; prev_end = None
for s, e, name in rows[a:b]:
    nm = name.replace("(anonymous namespace)::", "")[:44]
    print("%8.2f us  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, nm))
    prev_end = e
print("step span %.2f us" % ((rows[b][0] - t0) / 1e3))
