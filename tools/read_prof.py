"""Summarise a rocprofv3 rocpd sqlite output: kernel stats table and (if present) PMC counters per kernel."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
print("== top kernels (name, calls, total_us, avg_us, pct)")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-90s %6d %12.1f %10.2f %6.2f" % (r[0][:90], r[1], r[2]/1e3 if r[2] > 1e6 else r[2], r[3]/1e3 if r[3] > 1e5 else r[3], r[4]))
try:
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    if rows:
        print("== counters (kernel, counter, sum, dispatches)")
        for r in rows:
            print("%-60s %-28s %16.0f %5d" % (r[0][:60], r[1], r[2], r[3]))
except Exception as e:
    pass
