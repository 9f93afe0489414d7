"""Mask IoU of the chunked sequence run (sindslam_amd/sequence.py) against the sequential frame loop, per frame, for several warm-up lengths."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sindslam_amd.sequence import plan_chunks, process_sequence
from sindslam_amd.synth import SyntheticStream, TUM3
from sindslam_amd.dyna import DynaDetect
n, S, T = 66, 4, 2
bgr, depth = SyntheticStream(seed=4242).frames(0, n)
dd = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
ref = [None] + [dd.DetectDynaArea(bgr[f], depth[f], f)[0] for f in range(1, n)]
dd.close()
for W in (0, 4, 8, 16):
    got = process_sequence(bgr, depth, TUM3, streams=S, frames_per_step=T, warmup=W, want_keypoints=False)
    ious = []
    for f in range(1, n):
        a, r = got["dyna"][f] == 255, ref[f] == 255; u = np.logical_or(a, r).sum()
        ious.append(1.0 if u == 0 else float(np.logical_and(a, r).sum() / u))
    seams = [c.first for c in plan_chunks(n, S, W)[1:]]
    print(f"warm-up {W}: seams at {seams}; IoU per frame:", " ".join(f"{v:.3f}" for v in ious))
    print(f"   dynamic pixels in the reference per frame:", " ".join(str(int((ref[f] == 255).sum())) for f in range(1, n)))
