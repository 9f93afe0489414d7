"""The flow pyramid alone as the bench runs it: S slices of B pairs side by side (one FlowStage, stream and host thread each), nothing else on the GPU.
   python3 profiles/tools/flow_slices_alone.py [S] [B] [reps]"""
import sys, time, threading, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sindslam_amd.flow import FlowStage
from sindslam_amd._lib import lib

CAP = int(os.environ.get("SOLVER_WGS", "0"))
WAVE = os.environ.get("WAVE"); WAVE_ITEMS = int(os.environ.get("WAVE_ITEMS", "0")); WAVE_BANDS = int(os.environ.get("WAVE_BANDS", "0"))      # WAVE=0: k_sor_stream instead of k_sor_wave
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3; B = int(sys.argv[2]) if len(sys.argv) > 2 else 170; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w, h = 384, 288
rng = np.random.default_rng(3)
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
k = np.ones(5) / 5
img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, img))
i0 = np.stack([img[(b % 5):(b % 5) + h, (b % 3):(b % 3) + w] for b in range(B)]).astype(np.uint8)
i1 = np.stack([img[(b % 5) + 2:(b % 5) + 2 + h, (b % 3) + 3:(b % 3) + 3 + w] for b in range(B)]).astype(np.uint8)
import torch
stages = [FlowStage(w, h, B) for _ in range(S)]
if CAP:
    for st_ in stages:
        st_.set_solver_workgroups(CAP)
COEF = os.environ.get("COEF")
if COEF is not None:
    for st_ in stages:
        st_.set_coef_kernel(int(COEF))
if WAVE is not None:
    for st_ in stages:
        check_rc = lib().sind_flow_set_wave_solver(st_._h, int(WAVE), WAVE_ITEMS, WAVE_BANDS); assert check_rc == 0
d0 = torch.from_numpy(i0).cuda(); d1 = torch.from_numpy(i1).cuda()
outs = [(torch.empty((B, h, w), dtype=torch.float32, device="cuda"), torch.empty((B, h, w), dtype=torch.float32, device="cuda")) for _ in range(S)]
torch.cuda.synchronize()
def one(i):
    stages[i].deepflow_dev(d0.data_ptr(), d1.data_ptr(), B, outs[i][0].data_ptr(), outs[i][1].data_ptr()); stages[i].sync()
def round_():
    th = [threading.Thread(target=one, args=(i,)) for i in range(S)]
    for t in th: t.start()
    for t in th: t.join()
round_()
t0 = time.perf_counter()
for _ in range(reps): round_()
dt = (time.perf_counter() - t0) / reps
print(f"cap {CAP} wave {WAVE} items {WAVE_ITEMS} bands {WAVE_BANDS}: {S} slices x {B} pairs: {dt * 1e3:.1f} ms per round, {S * B / dt:.0f} pairs/s, mean |u| {float(outs[0][0].abs().mean()):.3f}")
