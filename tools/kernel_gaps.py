"""Idle time between consecutive kernels of a rocprofv3 rocpd trace (kernel-trace): where the timeline is not kernels.
usage: kernel_gaps.py <results.db> [name substring that marks the region of interest, default k_env_step]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
key = sys.argv[2] if len(sys.argv) > 2 else "k_env_step"
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
# steady state: from the middle of the trace on
first = [i for i, r in enumerate(rows) if key in r[0]]
if not first:
    raise SystemExit("no kernel matching " + key)
lo = first[len(first) // 2]; hi = first[-1]
busy = sum(r[2] - r[1] for r in rows[lo:hi]); span = rows[hi][1] - rows[lo][1]
gaps = {}
for a, b in zip(rows[lo:hi], rows[lo + 1:hi + 1]):
    g = b[1] - a[2]
    k = (a[0][:40], b[0][:40])
    c = gaps.setdefault(k, [0, 0]); c[0] += 1; c[1] += g
print("span %.1f us, kernels busy %.1f us (%.1f%%), %d kernels" % (span / 1e3, busy / 1e3, 100.0 * busy / span, hi - lo))
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%8.1f us total %6.2f us avg x%5d   %s -> %s" % (t / 1e3, t / 1e3 / n, n, k[0], k[1]))
