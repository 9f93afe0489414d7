"""Kernel trace of the in-order mode -> the k-means chain of one frame: python3 profiles/tools/km_chain_trace.py <rocprofv3 output dir>
(rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/tools/exact_mode_timing.py 97 32)"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
km = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_km_", "k_points", "k_labels", "k_depth_half"))]
# per-kernel totals
tot = collections.defaultdict(lambda: [0, 0])
for r in km:
    n = r["Kernel_Name"].split("(")[0].replace("sind::", "").replace("void ", ""); d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[n][0] += 1; tot[n][1] += d
print("k-means kernels over the run: calls, mean us")
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print(f"  {n:40s} {c:6d} {d / c / 1e3:8.1f}   total {d / 1e6:8.1f} ms")
# one chain in the middle of the run: from a k_depth_half to the next k_labels_to_u8
starts = [i for i, r in enumerate(km) if "k_depth_half" in r["Kernel_Name"]]
i0 = starts[len(starts) // 2 // 3 * 3]
chain = []; 
for r in km[i0:]:
    chain.append(r)
    if "k_labels_to_u8" in r["Kernel_Name"]: break
t0 = int(chain[0]["Start_Timestamp"]); prev_end = t0; busy = 0
print(f"\none frame's chain: {len(chain)} kernels")
for r in chain:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    print(f"  +{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  run {(e - s) / 1e3:7.1f}  {r['Kernel_Name'].split('(')[0].replace('sind::', '')[:40]}  grid {r.get('Grid_Size_X', '')} wg {r.get('Workgroup_Size_X', '')}")
    prev_end = e
print(f"chain wall {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us")
