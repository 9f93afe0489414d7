"""Where is the flow solver NOT running?  From a rocprofv3 --kernel-trace CSV of bench.py: per step (a step starts at its first k_gather_frames launch),
the union of the k_sor_fused intervals, the number of solver launches in flight over time, and the largest gaps with the kernels that ran inside them.
usage: python3 profiles/tools/solver_gaps.py <rocprof output dir>"""
import csv, glob, sys
from collections import Counter
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[0][0]
starts = [a for a, b, n in rows if 'k_gather_frames' in n]
steps = []
for a in starts:                                     # three slices launch a gather each: cluster launches closer than 50 ms
    if not steps or a - steps[-1] > 50e6: steps.append(a)
print('step starts (ms):', [round((a - t0) / 1e6, 1) for a in steps])
sor = sorted((a, b) for a, b, n in rows if 'k_sor_fused' in n)
un = []; cs, ce = sor[0]
for a, b in sor[1:]:
    if a > ce: un.append((cs, ce)); cs, ce = a, b
    else: ce = max(ce, b)
un.append((cs, ce))
for k in range(len(steps)):
    lo = steps[k]; hi = steps[k + 1] if k + 1 < len(steps) else rows[-1][1]
    busy = sum(min(b, hi) - max(a, lo) for a, b in un if b > lo and a < hi)
    # time-weighted number of solver launches in flight
    ev = []
    for a, b in sor:
        if b > lo and a < hi: ev.append((max(a, lo), 1)); ev.append((min(b, hi), -1))
    ev.sort(); depth = 0; last = lo; hist = Counter()
    for t, d in ev: hist[depth] += t - last; last = t; depth += d
    hist[depth] += hi - last
    print(f'step {k}: {(hi - lo) / 1e6:.1f} ms, solver busy {busy / 1e6:.1f} ms; in flight: ' + ', '.join(f'{d}: {v / 1e6:.1f} ms' for d, v in sorted(hist.items())))
    gaps = []
    prev = lo
    for a, b in un:
        if b <= lo or a >= hi: continue
        if a - prev > 0.3e6: gaps.append((prev, a))
        prev = max(prev, b)
    if hi - prev > 0.3e6: gaps.append((prev, hi))
    for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:6]:
        inside = Counter(); 
        for s, e, n in rows:
            if e > a and s < b: inside[n.split('(')[0].replace('void ', '').replace('sind::', '')[:28]] += min(e, b) - max(s, a)
        top = ', '.join(f'{n} {v / 1e6:.1f}' for n, v in inside.most_common(5))
        print(f'    gap {(a - lo) / 1e6:7.1f} .. {(b - lo) / 1e6:7.1f} ms ({(b - a) / 1e6:5.1f} ms): {top}')

# optional: python3 solver_gaps.py <dir> --timeline K  -> the first 70 ms of step K in 5-ms bins: busy time per kernel name inside each bin
if len(sys.argv) > 3 and sys.argv[2] == '--timeline':
    K = int(sys.argv[3]); lo = steps[K]
    for b in range(14):
        a0, a1 = lo + b * 5e6, lo + (b + 1) * 5e6
        inside = Counter()
        for s_, e_, n in rows:
            if e_ > a0 and s_ < a1: inside[n.split('(')[0].replace('void ', '').replace('sind::', '')[:26]] += min(e_, a1) - max(s_, a0)
        print(f'  {b * 5:3d}-{b * 5 + 5:3d} ms: ' + ', '.join(f'{n} {v / 1e6:.1f}' for n, v in inside.most_common(7)))
