"""Per flow slice (= HIP stream / queue that launches k_sor_stream): how much of the step is a flow kernel of that slice running, and how long the slice's stream had NOTHING
running between two of its kernels (dependency latency, launch starvation, or waiting for a free slot), from a rocprofv3 --kernel-trace csv.
usage: python3 profiles/tools/flow_stream_gaps.py <rocprof output dir>"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
byq = defaultdict(list)
for r in rows: byq[r[key]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
print('columns:', list(rows[0].keys()))
for q, ks in sorted(byq.items()):
    if not any('k_sor_stream' in n for _, _, n in ks): continue
    ks.sort(); span = ks[-1][1] - ks[0][0]; busy = sum(b - a for a, b, _ in ks)
    gaps = [ks[i + 1][0] - max(k[1] for k in ks[max(0, i - 3):i + 1]) for i in range(len(ks) - 1)]
    gaps = [g for g in gaps if g > 0]
    bins = [(0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)]
    print(f'{key} {q}: {len(ks)} kernels, span {span / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), gaps {sum(gaps) / 1e6:.1f} ms: ' +
          ', '.join(f'{lo / 1e3:.0f}-{hi / 1e3:.0f} us: {sum(1 for g in gaps if lo <= g < hi)} / {sum(g for g in gaps if lo <= g < hi) / 1e6:.1f} ms' for lo, hi in bins))
    tot = defaultdict(float); cnt = defaultdict(int)
    for a, b, n in ks: nm = n.split('(')[0].replace('void ', '').replace('sind::', '')[:28]; tot[nm] += b - a; cnt[nm] += 1
    print('    ' + ', '.join(f'{n} {cnt[n]} x {tot[n] / cnt[n] / 1e3:.0f} us' for n in sorted(tot, key=lambda n: -tot[n])[:6]))
