"""Duration distribution (us) of every non-flow kernel by grid size from a rocprofv3 --kernel-trace CSV, largest total first.
usage: python3 profiles/kernel_duration_dist.py <rocprof output dir>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if any(k in n for k in ('k_sor_fused', 'k_coef', 'k_resize_f32', 'k_warp_avg_iz', 'k_add_flow')):
        continue
    k = n.split('(')[0].replace('void ', '').replace('sind::', '')
    d.setdefault((k, int(r['Grid_Size_X'])), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v); print(k, len(v), 'p10 %.1f p50 %.1f p90 %.1f p99 %.1f mean %.1f sum_ms %.1f' % (np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.mean(), v.sum() / 1e3))
