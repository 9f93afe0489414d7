"""GPU busy time of a rocprofv3 --kernel-trace run (rocpd sqlite): union of the kernel intervals, mean concurrency, and the top kernels by summed duration inside [t0, t1] given as
fractions of the trace (default the middle 60 %: the timed steps of a bench run).  usage: db_busy.py <results.db> [lo hi]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2; hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
rows = db.execute("select name, start, end from kernels order by start").fetchall()
T0, T1 = rows[0][1], max(r[2] for r in rows); a = T0 + lo * (T1 - T0); b = T0 + hi * (T1 - T0)
sel = [(n, max(s, a), min(e, b)) for n, s, e in rows if e > a and s < b]
un = 0; cs = ce = None; tot = 0
for n, s, e in sel:
    tot += e - s
    if ce is None or s > ce:
        if ce is not None: un += ce - cs
        cs, ce = s, e
    else: ce = max(ce, e)
un += ce - cs
print(f"window {(b - a) / 1e6:.1f} ms: GPU busy (union) {un / 1e6:.1f} ms = {100 * un / (b - a):.1f} %, summed kernel time {tot / 1e6:.1f} ms, mean concurrency while busy {tot / un:.2f}, {len(sel)} launches")
acc = {}
for n, s, e in sel:
    k = n.split("(")[0].replace("void ", "").replace("sind::", "")[:56]; c = acc.setdefault(k, [0, 0]); c[0] += 1; c[1] += e - s
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {k:56s} {c:7d} launches {t / c / 1e3:8.2f} us avg {t / 1e6:8.2f} ms {100 * t / tot:5.1f} %")
