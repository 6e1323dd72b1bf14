"""Per-kernel duration table out of a rocprofv3 results .db (--kernel-trace): name, calls, avg / min / max in us.
usage: rocprof_db_stats.py results.db [substring]"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
     f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 6 desc")
sub = sys.argv[2] if len(sys.argv) > 2 else ""
print("%-70s %6s %10s %10s %10s %12s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us"))
for r in cur.execute(q):
    if sub in r[0]:
        print("%-70s %6d %10.1f %10.1f %10.1f %12.1f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
