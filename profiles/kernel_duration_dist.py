import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    for k in ('k_km_update', 'k_km_assign_partial', 'k_km_partial_dev', 'k_points', 'k_rag_stats', 'k_residual_mag', 'k_km_reset'):
        if k in n:
            d.setdefault((k, int(r['Grid_Size_X'])), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v = np.array(v); print(k, len(v), 'p10 %.1f p50 %.1f p90 %.1f p99 %.1f mean %.1f sum_ms %.1f' % (np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.mean(), v.sum() / 1e3))
