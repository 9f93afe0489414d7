"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd sqlite file (ROCm 7.2's default output): calls, total and average duration, share.
usage: python3 profiles/tools/db_kernel_stats.py <results.db> [runs the trace holds, to print per-run figures] [top N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); runs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end - start), min(start), max(end) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
span = max(r[4] for r in rows) - min(r[3] for r in rows)
print(f"{len(rows)} kernels, {n} launches ({n / runs:.0f} per run), summed duration {tot / 1e6:.2f} ms ({tot / 1e6 / runs:.3f} ms per run), first start to last end {span / 1e6:.1f} ms")
print(f"{'kernel':60s} {'calls/run':>10s} {'avg us':>9s} {'ms/run':>9s} {'share':>6s}")
for name, c, t, _, _ in rows[:top]:
    short = name.split("(")[0].replace("void ", "").replace("sind::", "")[:60]
    print(f"{short:60s} {c / runs:10.1f} {t / c / 1e3:9.2f} {t / 1e6 / runs:9.3f} {100.0 * t / tot:5.1f}%")
