"""Per-kernel table of a rocprofv3 --kernel-trace --stats run of bench.py (top kernels by summed duration): calls, average duration, share of the summed kernel time,
and -- where SURVEY 8d / DESIGN 3 give a per-unit figure -- the algorithmic bytes (or lane-instructions) per unit and the rate they imply.  Kernels of the three flow
slices and of the tails overlap on the GPU (mean concurrency ~2.7), so "rate" = units of all launches / SUMMED duration is what a launch achieves while it shares the
machine, not a device-level figure (that is bench.py's roofline object).
usage: python3 profiles/tools/kernel_table.py <kernel_stats.csv> <steps traced> <pairs per step> [width height]"""
import csv, sys

f, steps, pairs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (640, 480)
import numpy as np
w, h = int(np.float32(0.6) * W), int(np.float32(0.6) * H); px = []
while True:
    px.append(w * h)
    nw, nh = int(np.float32(w) * np.float32(0.95) + np.float32(0.5)), int(np.float32(h) * np.float32(0.95) + np.float32(0.5))
    if nw <= 25 or nh <= 25: break
    w, h = nw, nh
pyr = sum(px); big = sum(p for p in px if p > 8192); N = W * H
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# per frame pair and DeepFlow call (a pair runs 1 call, large-motion pairs 2; the refinement adds one level-0 pass of 5 x 5 iterations): bytes per unit x units
units = {
    "k_sor_stream": ("44 B / pixel update (SURVEY 8d; 40 B / pixel / launch compulsory)", 44.0 * big * 125),
    "k_sor_fused": ("44 B / pixel update, one-workgroup levels", 44.0 * (pyr - big) * 125),
    "k_coef_lanes": ("48 B / pixel / fixed-point iteration (6 planes read once, neighbours from lanes / registers; 6 written)", 48.0 * pyr * 5),
    "k_coef": ("48 B / pixel / fixed-point iteration (4 planes read, 6 written + stencil through L1/L2)", 48.0 * pyr * 5),
    "k_resize_f32_pair": ("16 B / destination pixel (two images or two flow components)", 16.0 * pyr * 2),
    "k_warp_avg_iz": ("20 B / pixel / level", 20.0 * pyr),
    "k_add_flow": ("24 B / pixel / level", 24.0 * pyr),
    "k_rag_stats": ("3 bit planes x C pieces + 2 bytes / pixel", (3 * 24 / 8 + 2.0) * N),
    "k_residual_mag": ("12 B / pixel", 12.0 * N), "k_mag_hist": ("5 B / pixel", 5.0 * N), "k_threshold_masks_dev": ("6 B / pixel", 6.0 * N),
    "k_km_seqsum": ("4 B / sample, sequential FP32 sums: 408 000 points x 3 coordinates x <= 4 passes", 4.0 * 408000 * 3 * 4),
    "k_km_compact": ("28 B / point / pass", 28.0 * 408000 * 4), "k_dilate_planes": ("2 x 24 bit planes / pixel", 6.0 * N),
}
print(f"{len(rows)} kernels, summed duration {tot / 1e6 / steps:.1f} ms per step ({steps} steps traced, {pairs} frame pairs per step)")
print(f"{'kernel':34s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>9s} {'share':>6s}  per-unit figure -> rate over the summed duration")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    name = r["Name"].split("(")[0].replace("void ", "").replace("sind::", ""); key = name.split("<")[0]
    t = float(r["TotalDurationNs"]); note = ""
    if key in units:
        what, per_pair = units[key]
        same = [q for q in rows if q["Name"].split("(")[0].replace("void ", "").replace("sind::", "").split("<")[0] == key]
        tk = sum(float(q["TotalDurationNs"]) for q in same)                      # all template instances of the kernel share the unit count
        note = f"{what}: {per_pair * pairs * steps / (tk * 1e-9) / 1e12:.2f} TB/s" + (" (all instances)" if len(same) > 1 else "")
    elif key == "k_peac_grow":
        note = "latency: one workgroup per frame, 140-520 BFS levels x ~4.9 us + 14 ns per seed"
    print(f"{name[:34]:34s} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f} {t / 1e6 / steps:9.1f} {100 * t / tot:5.1f}%  {note}")
