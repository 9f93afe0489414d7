"""Per-kernel sums of every counter in rocprofv3 --pmc output dirs (csv), for kernels whose name contains <substring>:
   python3 profiles/tools/pmc_table.py <substring> <dir> [...]"""
import csv, glob, sys
from collections import defaultdict
kernel = sys.argv[1]
for d in sys.argv[2:]:
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs:
        print(d, 'no counter file'); continue
    tot = defaultdict(float); n = defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        if kernel in r['Kernel_Name']:
            key = (r['Counter_Name'], r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')))
            tot[key] += float(r['Counter_Value']); n[key] += 1
    for key in sorted(tot):
        print(f'{d.split("/")[-1]}: {kernel} wg={key[1]} {key[0]}: launches {n[key]} sum {tot[key]:.4g} per launch {tot[key] / n[key]:.4g}')
