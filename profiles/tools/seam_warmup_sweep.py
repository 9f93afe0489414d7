"""Chunked sequence mode against the in-order mode (both on the GPU) for several chunk warm-up lengths: how many owned frames of the later chunks fall below
IoU 0.99, where they sit relative to their chunk's first owned frame, and how many dynamic pixels they have.
usage: python3 profiles/tools/seam_warmup_sweep.py [frames] [chunks]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sindslam_amd.sequence import plan_chunks, process_sequence, process_sequence_exact
from sindslam_amd.synth import SyntheticStream, TUM3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 481; S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base_b, base_d = SyntheticStream(seed=12345).frames(0, 50)
idx = [i if i < 50 else 98 - i for i in (np.arange(n) % 98)]                # the bench's ping-pong over 50 generated frames
bgr, depth = base_b[idx], base_d[idx]
ref = process_sequence_exact(bgr, depth, TUM3, frames_per_step=32, want_keypoints=False)["dyna"]
for W in (8, 16, 24, 32, 48, 64):
    got = process_sequence(bgr, depth, TUM3, streams=S, frames_per_step=4, warmup=W, want_keypoints=False)["dyna"]
    chunks = plan_chunks(n, S, W)
    bad = []; ious = []
    for c in chunks[1:]:
        for f in range(c.first, c.last):
            a, r = got[f] == 255, ref[f] == 255; u = np.logical_or(a, r).sum(); v = 1.0 if u == 0 else float(np.logical_and(a, r).sum() / u)
            ious.append(v)
            if v < 0.99: bad.append((f - c.first, round(v, 3), int(r.sum())))
    ious = np.array(ious)
    print(f"warm-up {W:2d}: {len(ious)} owned frames of chunks 1.., mean {ious.mean():.4f} min {ious.min():.3f}, below 0.99: {len(bad)}  (offset in chunk, IoU, dynamic pixels of the in-order mask): {bad[:14]}")
