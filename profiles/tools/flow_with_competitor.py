"""How much do MANY SMALL kernels on high-priority streams cost the flow pyramid?  Three flow slices of 171 pairs (as in flow_slices_alone.py) while N host threads launch
per-frame-sized elementwise kernels (640 x 480, one torch op = one kernel) on high-priority streams at a fixed total rate -- a stand-in for the tails' per-frame kernels
(6 kernels x 512 frames per 350 ms step = 8.8 k kernels/s).
   python3 profiles/tools/flow_with_competitor.py [kernels per second] [threads] [elements per kernel]"""
import sys, time, threading, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sindslam_amd.flow import FlowStage
rate = float(sys.argv[1]) if len(sys.argv) > 1 else 8800.0; NT = int(sys.argv[2]) if len(sys.argv) > 2 else 8; NE = int(sys.argv[3]) if len(sys.argv) > 3 else 307200
S, B, w, h = 3, 171, 384, 288
rng = np.random.default_rng(3)
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
i0 = np.stack([img[(b % 5):(b % 5) + h, (b % 3):(b % 3) + w] for b in range(B)]).astype(np.uint8)
i1 = np.stack([img[(b % 5) + 2:(b % 5) + 2 + h, (b % 3) + 3:(b % 3) + 3 + w] for b in range(B)]).astype(np.uint8)
stages = [FlowStage(w, h, B) for _ in range(S)]
d0 = torch.from_numpy(i0).cuda(); d1 = torch.from_numpy(i1).cuda()
outs = [(torch.empty((B, h, w), dtype=torch.float32, device="cuda"), torch.empty((B, h, w), dtype=torch.float32, device="cuda")) for _ in range(S)]
torch.cuda.synchronize()
stop = False; launched = [0] * NT
def competitor(k):
    st = torch.cuda.Stream(priority=-1); a = torch.rand(NE, device="cuda"); b = torch.rand(NE, device="cuda"); c = torch.empty_like(a)
    per = NT / rate; nxt = time.perf_counter()
    with torch.cuda.stream(st):
        while not stop:
            torch.add(a, b, out=c); launched[k] += 1
            nxt += per
            d = nxt - time.perf_counter()
            if d > 0: time.sleep(d)
            if launched[k] % 64 == 0: st.synchronize()
def one(i):
    stages[i].deepflow_dev(d0.data_ptr(), d1.data_ptr(), B, outs[i][0].data_ptr(), outs[i][1].data_ptr()); stages[i].sync()
def round_():
    th = [threading.Thread(target=one, args=(i,)) for i in range(S)]
    for t in th: t.start()
    for t in th: t.join()
round_()
t0 = time.perf_counter()
for _ in range(3): round_()
alone = (time.perf_counter() - t0) / 3
ct = [threading.Thread(target=competitor, args=(k,)) for k in range(NT)] if rate > 0 else []
for t in ct: t.start()
time.sleep(0.2); n0 = sum(launched); t0 = time.perf_counter()
for _ in range(3): round_()
dt = (time.perf_counter() - t0) / 3; n1 = sum(launched)
stop = True
for t in ct: t.join()
print(f"flow alone {alone * 1e3:.1f} ms per round; with {(n1 - n0) / (3 * dt):.0f} kernels/s of {NE} elements on {NT} high-priority streams: {dt * 1e3:.1f} ms per round")
