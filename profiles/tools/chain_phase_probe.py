"""k_coarse_chain by phase: one level of a given size with (fixed-point iterations, SOR iterations) varied, so that the kernel trace separates the fixed cost of a level,
of a fixed-point iteration (coefficients + solver prologue / epilogue) and of a SOR iteration.  Run under rocprofv3 --kernel-trace; read with db_kernel_list.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sindslam_amd.flow import FlowStage
fs = FlowStage(384, 288, 2)
rng = np.random.default_rng(0)
for (w, h) in [(33, 26), (45, 34), (52, 39), (64, 48), (73, 55)]:
    i0 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); i1 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); z = np.zeros((1, h, w), np.float32)
    for fp, sor in [(1, 1), (1, 25), (1, 50), (5, 25), (5, 1)]:
        for r in range(3):
            fs.varref_f32(i0, i1, z, z, fp, sor, 4.0, 0.5 / 3, 5.0 / 3, 1.6)
fs.close()
