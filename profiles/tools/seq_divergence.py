"""Where do the GPU loop and the oracle loop part on a long sequence?  Per frame: mask IoU, k-means label mismatch, merged-label mismatch,
homography / threshold equality, piece count.  Run on the GPU box:  python profiles/tools/seq_divergence.py [frames] [seed] [chunks]
With `chunks` > 0 the same sequence also goes through the chunked (throughput) mode -- that many chunks, 24 state warm-up frames each -- and its masks
are compared with the oracle's sequential masks: mean / minimum IoU and the number of frames below 0.99."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from sindslam_amd.dyna import DynaDetect
from sindslam_amd.synth import SyntheticStream, TUM3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
bgr, depth = SyntheticStream(seed=seed).frames(0, n)
K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ref = O.DynaDetect(bgr[0], bgr[0].copy(), *K); gpu = DynaDetect(bgr[0], bgr[0].copy(), *K)
ref_dyn = np.zeros((n,) + bgr.shape[1:3], bool); all_eq = 0; worst = 1.0
for f in range(1, n):
    rd, rl = ref.detect(bgr[f], depth[f]); r = ref.debug()
    gd, gl = gpu.DetectDynaArea(bgr[f], depth[f], f); g = gpu.debug()
    u = np.logical_or(gd == 255, rd == 255).sum(); iou = 1.0 if u == 0 else np.logical_and(gd == 255, rd == 255).sum() / u
    ff = np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]])
    print(f"frame {f:3d} IoU {iou:.4f} dyn_px {int((rd == 255).sum()):6d} | flow_eq {np.array_equal(g['flow_full'], ff)} km_mismatch {(g['kmeans_label'] != r['kmeans_label']).mean():.2e} "
          f"label_mismatch {(gl != rl).mean():.2e} pieces {g['info'][2]}/{r['info'][2]} pairs {g['info'][1]}/{r['info'][1]} H_eq {np.array_equal(g['H'], r['H'])} "
          f"thr_eq {np.array_equal(g['thr'], r['thr'])} low_eq {np.array_equal(g['mask_low'], r['mask_low'])} high_eq {np.array_equal(g['mask_high'], r['mask_high'])} "
          f"occ1_eq {np.array_equal(g['occ1'], r['occ1'])} ctr_maxdiff {np.abs(g['centers'] - r['centers']).max():.2e}", flush=True)
    ref_dyn[f] = rd == 255; worst = min(worst, iou)
    all_eq += int(np.array_equal(gd, rd) and np.array_equal(gl, rl) and np.array_equal(g['flow_full'], ff) and np.array_equal(g['H'], r['H']) and np.array_equal(g['kmeans_label'], r['kmeans_label']))
print(f"SUMMARY in-order: {n - 1} frames, imgDyna + imgLabel + flow + homography + k-means labels all equal on {all_eq} frames, minimum mask IoU {worst:.4f}", flush=True)
if chunks > 0:
    from sindslam_amd.sequence import process_sequence
    del gpu
    out = process_sequence(bgr, depth, TUM3, streams=chunks, frames_per_step=4, warmup=24, want_keypoints=False)
    ious = []
    for f in range(1, n):
        a = out["dyna"][f] == 255; b = ref_dyn[f]; u = np.logical_or(a, b).sum()
        ious.append(1.0 if u == 0 else float(np.logical_and(a, b).sum() / u))
    ious = np.array(ious)
    print(f"SUMMARY chunked: {chunks} chunks x 24 warm-up frames vs the oracle's sequential masks: mean IoU {ious.mean():.5f}, minimum {ious.min():.4f}, frames below 0.99: {(ious < 0.99).sum()} of {len(ious)}, below 0.97: {(ious < 0.97).sum()}", flush=True)
