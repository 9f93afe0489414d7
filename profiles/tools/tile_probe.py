import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sindslam_amd.flow import FlowStage
fs = FlowStage(384, 288, 2)
rng = np.random.default_rng(0)
for (w, h) in [(128, 96), (384, 288)]:
    i0 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); i1 = rng.uniform(0, 255, (1, h, w)).astype(np.float32); z = np.zeros((1, h, w), np.float32)
    for sor in (1, 5, 9, 13):
        for r in range(3):
            fs.varref_f32(i0, i1, z, z, 1, sor, 4.0, 0.5 / 3, 5.0 / 3, 1.6)
fs.close()
