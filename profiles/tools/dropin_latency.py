"""The reference's call pattern (rgbd_tum_noros.cc:131-139, Frame.cc:308): ONE frame at a time, host pointers, in order, through the two classes --
sind_dyna_detect + sind_dyna_dilate15 + sind_orb_extract at B = 1.  Prints per-call latencies over N frames and the flow stage alone at B = 1 / 8 / 32.
usage: python3 profiles/tools/dropin_latency.py [frames] [--flow-only] [--batches 1,8,32]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sindslam_amd.dyna import DynaDetect
from sindslam_amd.orb import ORBextractor
from sindslam_amd.flow import FlowStage
from sindslam_amd.synth import SyntheticStream, TUM3

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 120
P = 40
HD = "--720p" in sys.argv                # BASELINE.json config 5: 1280 x 720, D455 intrinsics x 2, 3-level flow pyramid
if HD:
    from sindslam_amd.synth import D455
    TUM3 = dict(D455); TUM3.update(fx=D455["fx"] * 2, fy=D455["fy"] * 2, cx=D455["cx"] * 2, cy=D455["cy"] * 2)
    bgr, depth = SyntheticStream(1280, 720, seed=12345, intr=TUM3).frames(0, P)
else:
    bgr, depth = SyntheticStream(seed=12345).frames(0, P)
pp = lambda f: (f % (2 * P - 2)) if (f % (2 * P - 2)) < P else 2 * P - 2 - (f % (2 * P - 2))
gray = [((b[..., 0].astype(np.int32) * 4899 + b[..., 1].astype(np.int32) * 9617 + b[..., 2].astype(np.int32) * 1868 + 8192) >> 14).astype(np.uint8) for b in bgr]   # Tracking.cc:246 (caller side)
out = {}
if "--flow-only" not in sys.argv:
    gpu = DynaDetect(bgr[1], bgr[0], TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"], debug=False, overlap="--no-overlap" not in sys.argv)
    if HD:
        gpu.set_flow_max_levels(3)
    orb = ORBextractor(1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"])
    td, tm, to = [], [], []
    for f in range(2, 2 + n + 10):
        k = pp(f)
        t0 = time.perf_counter(); dyna, label = gpu.DetectDynaArea(bgr[k], depth[k], f)
        t1 = time.perf_counter(); mask = gpu.dilate15(dyna)
        t2 = time.perf_counter(); kps, desc = orb(gray[k], mask)
        t3 = time.perf_counter()
        if f == 11:
            gpu.timing(True); gpu.timing_fine(True)
        if f >= 12:
            td.append(t1 - t0); tm.append(t2 - t1); to.append(t3 - t2)
    tot = np.array(td) + np.array(tm) + np.array(to)
    out["dropin"] = {"frames": len(td), "ms_per_frame": float(tot.mean() * 1e3), "fps": float(1.0 / tot.mean()), "detect_ms": float(np.mean(td) * 1e3), "dilate15_ms": float(np.mean(tm) * 1e3),
                     "orb_ms": float(np.mean(to) * 1e3), "detect_ms_p50": float(np.median(td) * 1e3), "detect_ms_max": float(np.max(td) * 1e3)}
    out["dropin"]["detect_stages_ms"] = gpu.timing()
    tf = gpu.timing_fine(); nm = max(len(td), 1)
    names = {0: "occ gpu+d2h", 1: "occ pack", 2: "occ end points", 3: "occ peac (host parts + grow)", 4: "occ contour filter", 5: "occ close", 6: "seg pieces", 8: "seg rag (gpu)", 9: "seg merge", 10: "seg sort+paint+pack",
             12: "pieces open", 13: "pieces contours", 14: "pieces masks", 15: "pieces lianjie", 16: "pieces centre", 20: "flow weights", 21: "flow sort+wait", 22: "flow homography", 23: "flow pack", 24: "fusion low", 25: "fusion clusters", 26: "fusion fill", 27: "fusion out+state"}
    out["dropin"]["tail_fine_ms"] = {names.get(i, str(i)): round(v / nm, 3) for i, v in enumerate(tf) if v > 0}
    gpu.close(); orb.close()
# the flow stage alone: DeepFlow of B pairs (host in / out included) and on device pointers
fw, fh = 384, 288
rng = np.random.default_rng(1)
import torch
BS = [int(x) for x in sys.argv[sys.argv.index("--batches") + 1].split(",")] if "--batches" in sys.argv else [1, 8, 32]
for B in BS:
    fs = FlowStage(fw, fh, B)
    from sindslam_amd._lib import lib
    g = np.stack([np.ascontiguousarray(gray[pp(i)][:fh, :fw]) for i in range(B + 2)])
    i0 = torch.from_numpy(g[2:]).cuda(); i1 = torch.from_numpy(g[:-2]).cuda(); u = torch.empty((B, fh, fw), dtype=torch.float32, device="cuda"); v = torch.empty_like(u)
    torch.cuda.synchronize()
    for r in range(3):
        fs.deepflow_dev(i0.data_ptr(), i1.data_ptr(), B, u.data_ptr(), v.data_ptr()); fs.sync()
    t0 = time.perf_counter(); R = 10
    for r in range(R):
        fs.deepflow_dev(i0.data_ptr(), i1.data_ptr(), B, u.data_ptr(), v.data_ptr()); fs.sync()
    dt = (time.perf_counter() - t0) / R
    out[f"deepflow_B{B}_ms"] = dt * 1e3
    fs.close()
print(json.dumps(out))
