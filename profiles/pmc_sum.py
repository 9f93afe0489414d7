"""Sums one rocprofv3 --pmc counter over the launches of one kernel (per-dispatch CSV of `--output-format csv`).
usage: python3 profiles/pmc_sum.py <kernel substring> <rocprof output dir> [...]"""
import csv
import glob
import sys

kernel = sys.argv[1]
for d in sys.argv[2:]:
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot, n, name = 0.0, 0, ''
    for r in csv.DictReader(open(f)):
        if kernel in r['Kernel_Name']:
            tot += float(r['Counter_Value']); n += 1; name = r['Counter_Name']
    print(f'{d}: {kernel} launches {n} {name} sum {tot:.1f} per launch {tot / max(n, 1):.2f}')
