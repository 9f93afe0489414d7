"""In-order ("exact") sequence mode on one GPU: frames/s and, with SIND_TAIL_TIMING=1, the per-frame stage times of the two tail chains.
usage: SIND_TAIL_TIMING=1 python3 profiles/tools/exact_mode_timing.py [frames] [frames_per_step]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sindslam_amd.sequence import process_sequence_exact
from sindslam_amd.synth import SyntheticStream, TUM3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 130; T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
base_b, base_d = SyntheticStream(seed=12345).frames(0, 34)
idx = [i if i < 34 else 66 - i for i in (np.arange(n) % 66)]            # ping-pong over 34 generated frames
bgr, depth = base_b[idx], base_d[idx]
process_sequence_exact(bgr[:T + 1], depth[:T + 1], TUM3, frames_per_step=T, want_keypoints=False)      # warm-up (library, allocations)
t0 = time.perf_counter()
process_sequence_exact(bgr, depth, TUM3, frames_per_step=T, want_keypoints=False)
dt = time.perf_counter() - t0
print(f"in-order mode: {n - 1} frames in {dt:.3f} s = {(n - 1) / dt:.1f} frames/s (frames_per_step {T}; includes pipeline creation and the host-side uploads)")
