"""Launch durations of one kernel by grid size (rocprofv3 --kernel-trace rocpd sqlite): python3 profiles/tools/db_kernel_by_grid.py <results.db> <kernel substring>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = next(c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size"))
wx = next(c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size"))
rows = db.execute(f"select {gx}, {wx}, count(*), avg(end - start), min(end - start), max(end - start) from kernels where name like ? group by {gx}, {wx} order by {gx} desc", (f"%{sub}%",)).fetchall()
print(f"{'grid':>9s} {'wg':>5s} {'workgroups':>10s} {'launches':>8s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for g, w, n, a, mn, mx in rows:
    print(f"{g:9d} {w:5d} {g // w:10d} {n:8d} {a / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}")
