"""How much of a rocprofv3 kernel trace has two or more kernels in flight (streams overlapping)?  usage: kernel_overlap.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
ev = []
for n, s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; t = {0: 0, 1: 0, 2: 0}
for x, d in ev:
    t[min(depth, 2)] += x - last; last = x; depth += d
tot = sum(t.values())
print("idle %.1f%%  one kernel %.1f%%  two or more %.1f%%  (span %.1f ms)" % (100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, tot / 1e6))
# which kernels ran inside an env step?
envs = [(s, e) for n, s, e in rows if "k_env_step" in n]
inside = {}
j = 0
for n, s, e in rows:
    if "k_env_step" in n: continue
    while j < len(envs) and envs[j][1] < s: j += 1
    if j < len(envs) and envs[j][0] <= s < envs[j][1]:
        c = inside.setdefault(n[:60], [0, 0]); c[0] += 1; c[1] += e - s
for k, (c, d) in sorted(inside.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  started inside an env step: %5d x %8.1f us avg  %s" % (c, d / c / 1e3, k))
