"""Is the GPU ever idle inside the timed steps?  From a rocprofv3 --kernel-trace CSV of bench.py: union of ALL kernels' execution intervals against the span of the
timed steps, the idle gaps by length, and which kernels run on both sides of the long ones.
usage: python3 profiles/tools/gpu_idle.py <rocprof output dir> [skip_first_fraction]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sind::", "").replace("void ", "")) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip          # skip warm-up / set-up
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
busy = 0; cs, ce, cn = rows[0]; gaps = []
for a, b, n in rows[1:]:
    if a > ce:
        busy += ce - cs; gaps.append((a - ce, cn, n)); cs, ce, cn = a, b, n
    else:
        if b > ce: ce, cn = b, n
busy += ce - cs
print(f"span {span / 1e6:.1f} ms, some kernel running {busy / 1e6:.1f} ms = {busy / span:.3f}; idle {(span - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(g):6d}, {sum(x[0] for x in g) / 1e6:8.2f} ms")
for d, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {d / 1e3:9.1f} us between {a[:36]:36s} and {b[:36]}")
