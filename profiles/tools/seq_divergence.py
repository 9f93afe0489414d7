"""Where do the GPU loop and the oracle loop part on a long sequence?  Per frame: mask IoU, k-means label mismatch, merged-label mismatch,
homography / threshold equality, piece count.  Run on the GPU box:  python profiles/tools/seq_divergence.py [frames] [seed]"""
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
ref = O.DynaDetect(bgr[0], bgr[0].copy(), *K); gpu = DynaDetect(bgr[0], bgr[0].copy(), *K)
for f in range(1, n):
    rd, rl = ref.detect(bgr[f], depth[f]); r = ref.debug()
    gd, gl = gpu.DetectDynaArea(bgr[f], depth[f], f); g = gpu.debug()
    u = np.logical_or(gd == 255, rd == 255).sum(); iou = 1.0 if u == 0 else np.logical_and(gd == 255, rd == 255).sum() / u
    ff = np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]])
    print(f"frame {f:3d} IoU {iou:.4f} dyn_px {int((rd == 255).sum()):6d} | flow_eq {np.array_equal(g['flow_full'], ff)} km_mismatch {(g['kmeans_label'] != r['kmeans_label']).mean():.2e} "
          f"label_mismatch {(gl != rl).mean():.2e} pieces {g['info'][2]}/{r['info'][2]} pairs {g['info'][1]}/{r['info'][1]} H_eq {np.array_equal(g['H'], r['H'])} "
          f"thr_eq {np.array_equal(g['thr'], r['thr'])} low_eq {np.array_equal(g['mask_low'], r['mask_low'])} high_eq {np.array_equal(g['mask_high'], r['mask_high'])} "
          f"occ1_eq {np.array_equal(g['occ1'], r['occ1'])} ctr_maxdiff {np.abs(g['centers'] - r['centers']).max():.2e}", flush=True)
