"""DeepFlow of B pairs alone (device pointers): the default solver choice at that batch size against the one-wave pipelines on every level beyond one workgroup (mode 6).
usage: python3 profiles/tools/wave_small_b.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from sindslam_amd.flow import FlowStage
w, h = 384, 288
rng = np.random.default_rng(3)
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
k = np.ones(5) / 5
img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, img))
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 8, 16, 32, 64]:
    i0 = np.stack([img[(b % 5):(b % 5) + h, (b % 3):(b % 3) + w] for b in range(B)]).astype(np.uint8)
    i1 = np.stack([img[(b % 5) + 2:(b % 5) + 2 + h, (b % 3) + 3:(b % 3) + 3 + w] for b in range(B)]).astype(np.uint8)
    d0 = torch.from_numpy(i0).cuda(); d1 = torch.from_numpy(i1).cuda()
    u = torch.empty((B, h, w), dtype=torch.float32, device="cuda"); v = torch.empty_like(u)
    fs = FlowStage(w, h, B); res = {}; ref = None
    for name, mode, items in (("default", 4, 0), ("wave/2048", 6, 2048), ("wave/1024", 6, 1024), ("wave/512", 6, 512)):
        fs.set_sor_variant(mode, 5, 64, 64); fs.set_wave_solver(True, items, 0)
        fs.deepflow_dev(d0.data_ptr(), d1.data_ptr(), B, u.data_ptr(), v.data_ptr()); fs.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fs.deepflow_dev(d0.data_ptr(), d1.data_ptr(), B, u.data_ptr(), v.data_ptr())
        fs.sync(); res[name] = (time.perf_counter() - t0) / 5 * 1e3
        cur = (u.clone(), v.clone())
        if ref is None: ref = cur
        else: assert torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1]), name
    print(f"B={B:3d}: " + "  ".join(f"{n} {t:7.2f} ms" for n, t in res.items()), flush=True)
    fs.close()
