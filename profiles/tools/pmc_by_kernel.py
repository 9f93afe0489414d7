"""Sum of every collected counter per kernel name over a rocprofv3 --pmc run (csv): python3 profiles/tools/pmc_by_kernel.py <dir> [top]"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/*/*counter_collection.csv')[0]
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); names = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sind::', '')[:34]
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); names.add(r['Counter_Name'])
    if r['Counter_Name'] == sorted(names)[0]: n[k] += 1
names = sorted(names)
print(f"{'kernel':34s} {'launches':>8s} " + ' '.join(f'{c:>20s}' for c in names))
key = 'SQ_WAVE_CYCLES' if 'SQ_WAVE_CYCLES' in names else names[0]
allsum = {c: sum(tot[k][c] for k in tot) for c in names}
for k in sorted(tot, key=lambda k: -tot[k][key])[:top]:
    print(f"{k:34s} {n[k]:8d} " + ' '.join(f'{tot[k][c]:13.4g} ({100 * tot[k][c] / allsum[c]:4.1f}%)' for c in names))
print(f"{'all kernels':34s} {sum(n.values()):8d} " + ' '.join(f'{allsum[c]:20.4g}' for c in names))
