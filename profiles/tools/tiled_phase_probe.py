"""k_sor_fused (tiled levels) by iterations per launch: one level, fuse = 1, 3, 5, 7 -> the kernel trace separates a launch's fixed cost from its iterations.
Run under rocprofv3 --kernel-trace; read with db_kernel_list.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sindslam_amd.flow import FlowStage
fs = FlowStage(384, 288, 2)
rng = np.random.default_rng(0)
for (w, h) in [(100, 80), (128, 96), (384, 288)]:
    i0 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); i1 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); z = np.zeros((1, h, w), np.float32)
    for fuse in (1, 3, 5, 7):
        fs.set_latency_tiles(False); fs.set_sor_variant(4, fuse, 64, 64)
        fs.varref_f32(i0, i1, z, z, 1, 21, 4.0, 0.5 / 3, 5.0 / 3, 1.6)
fs.close()
