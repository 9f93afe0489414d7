"""The flow solver alone: DeepFlow on B textured pairs through the stage handle (no tails, no ORB), for PMC passes and A/B timing.
   python3 profiles/tools/sor_only.py [B] [reps] [w] [h]"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sindslam_amd.flow import FlowStage

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = int(sys.argv[3]) if len(sys.argv) > 3 else 640; h = int(sys.argv[4]) if len(sys.argv) > 4 else 480
rng = np.random.default_rng(3)
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
k = np.ones(5) / 5
img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, img))
i0 = np.stack([img[(b % 5):(b % 5) + h, (b % 3):(b % 3) + w] for b in range(B)]).astype(np.uint8)
i1 = np.stack([img[(b % 5) + 2:(b % 5) + 2 + h, (b % 3) + 3:(b % 3) + 3 + w] for b in range(B)]).astype(np.uint8)
fs = FlowStage(w, h, B)
fs.deepflow(i0, i1)
t0 = time.perf_counter()
for _ in range(reps):
    u, v = fs.deepflow(i0, i1)
dt = (time.perf_counter() - t0) / reps
print(f"B={B} {w}x{h}: {dt * 1e3:.1f} ms per batch, {dt * 1e3 / B:.2f} ms per pair, mean |u| {np.abs(u).mean():.3f} (host copies included)")
