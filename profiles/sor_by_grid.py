"""Bins the launches of one kernel (default k_sor_fused) of a rocprofv3 --kernel-trace CSV by grid size (= pyramid level): count, total
and mean duration.
usage: python3 profiles/sor_by_grid.py <rocprof output dir> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
bins = defaultdict(lambda: [0, 0.0]); tot = 0.0
for r in csv.DictReader(open(f)):
    if (sys.argv[2] if len(sys.argv) > 2 else 'k_sor_fused') not in r['Kernel_Name']:
        continue
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Workgroup_Size_X']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    bins[g][0] += 1; bins[g][1] += d; tot += d
print('workgroups_x images threads : launches total_us mean_us share')
for g, (n, d) in sorted(bins.items(), key=lambda kv: -kv[1][1]):
    print(f'{g[0]:5d} {g[1]:4d} {g[2]:5d} : {n:5d} {d:10.0f} {d / n:8.1f} {100 * d / tot:5.1f}%')
print('total_us', tot)
