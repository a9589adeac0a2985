"""Summarise rocprofv3 (rocpd sqlite) output of tools/rocprof_passes.sh into a small text report."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
lines = []
for db in sorted(glob.glob(os.path.join(out, "*", "*_results.db"))):
    con = sqlite3.connect(db)
    cur = con.cursor()
    name = os.path.basename(os.path.dirname(db))
    if name == "trace":
        lines.append("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
        lines.append(f"{'calls':>6} {'avg_us':>10} {'total_us':>12} {'pct':>6}  kernel")
        # from the dispatch table itself (durations in ns there), always printed in microseconds: a 25 ms kernel reads
        # 25420.00 under avg_us (round 4's table took the stats view's numbers and guessed their unit by size)
        rows = list(cur.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name "
                                "order by sum(duration) desc"))
        whole = sum(r[2] for r in rows) or 1
        for n, calls, tot, avg in rows[:12]:
            lines.append(f"{calls:6d} {avg / 1e3:10.2f} {tot / 1e3:12.2f} {100.0 * tot / whole:6.2f}  {n[:110]}")
        for r in cur.execute("select name, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, grid_x, workgroup_x, "
                             "avg(duration), count(*) from kernels where name like '%mz::%' group by name"):
            lines.append(f"   {r[0][:70]}: lds={r[1]} vgpr={r[2]} agpr={r[3]} sgpr={r[4]} grid={r[5]} wg={r[6]} "
                         f"avg_ns={r[7]:.0f} n={r[8]}")
    else:
        lines.append(f"== {name} (rocprofv3 --pmc), per-dispatch averages ==")
        for k, c, v, n in cur.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                "where kernel_name like '%mz::%' group by kernel_name, counter_name"):
            lines.append(f"   {k[k.index('mz::') + 4:][:51]:52s} {c:28s} {v:16.1f}  (n={n})")
text = "\n".join(lines)
print(text)
with open(os.path.join(out, "summary.txt"), "w") as f:
    f.write(text + "\n")
