"""How busy is the GPU during the tails?  From a rocprofv3 --kernel-trace CSV: union and sum of the execution intervals of the tail
kernels (everything except the flow / ORB kernels) and of the flow kernels.
usage: python3 profiles/tail_gpu_busy.py <rocprof output dir> <steps incl. warm-up>"""
import csv
import glob
import sys

FLOW = ("k_sor_stream", "k_sor_fused", "k_sor_color", "k_coef", "k_add_flow", "k_resize_f32", "k_warp_avg_iz", "k_resize_u8", "k_u8_to_f32_blur3", "k_bgr2gray", "k_mag_max", "k_scale")
ORB = ("k_fast_cells", "k_pad_reflect101", "k_blur7", "k_ic_angle", "k_brief", "k_compact_cells", "k_copy_into_slab")
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]; steps = int(sys.argv[2])
iv = {"flow": [], "tail": []}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if any(k in n for k in ORB):
        continue
    iv["flow" if any(k in n for k in FLOW) else "tail"].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k, v in iv.items():
    v.sort(); un = 0; cs, ce = v[0]
    for a, b in v[1:]:
        if a > ce:
            un += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    un += ce - cs
    sm = sum(b - a for a, b in v)
    print(f'{k}: {len(v) / steps:.0f} kernels/step, busy (union) {un / 1e6 / steps:.1f} ms/step, summed {sm / 1e6 / steps:.1f} ms/step, mean concurrency {sm / un:.2f}')
