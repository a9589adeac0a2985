"""Where does the time between kernels go?  Reads a rocprofv3 rocpd database and prints, for the busiest
stretch of the trace, kernel time vs wall span and the largest inter-kernel gaps by (previous, next) kernel."""
import glob
import sqlite3
import sys
from collections import defaultdict

db = glob.glob(sys.argv[1] + "/*/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[len(rows) // 2:]  # second half: the graph-mode acts
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  ({100 * busy / span:.1f} %)")
gaps = defaultdict(lambda: [0, 0])
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = gaps[(n0[:40], n1[:40])]
    g[0] += max(0, s1 - e0)
    g[1] += 1
for k, (tot, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"{tot / 1e6:8.2f} ms  n={n:5d}  avg {tot / n / 1e3:7.1f} us   {k[0]}  ->  {k[1]}")
idx = [i for i, r in enumerate(rows) if 'select_kernel' in r[0]]
mid = idx[-6] - 2 if len(idx) > 6 else len(rows) - 400
print("--- a stretch of the trace (gap before, duration, kernel) ---")
for (n0, s0, e0), (n1, s1, e1) in list(zip(rows, rows[1:]))[mid:mid + 36]:
    print(f"{(s1 - e0) / 1e3:8.1f} {(e1 - s1) / 1e3:8.1f}  {n1[:90]}")
