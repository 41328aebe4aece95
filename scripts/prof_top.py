"""Print the rocprofv3 kernel summary (top_kernels view of the rocpd sqlite output) as CSV."""
import glob, sqlite3, sys
path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
print("kernel,calls,total_us,avg_us,percent")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.2f' % (r[0].replace('(anonymous namespace)::', '')[:90], r[1], r[2], r[3], r[4]))
