"""Lab builds only (make -C sindslam_amd/csrc lab): where k_sor_stream's step cycles go, per wave -- the solver alone on B pairs.
   python3 profiles/tools/ss_step_profile.py [B] [w] [h]"""
import sys, os, ctypes, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sindslam_amd.flow import FlowStage
from sindslam_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 170
w = int(sys.argv[2]) if len(sys.argv) > 2 else 640; h = int(sys.argv[3]) if len(sys.argv) > 3 else 480
lib = _lib.lib()
if not lib.sind_lab_build():
    raise SystemExit("not a lab build")
lib.sind_lab_ss_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(3)
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h + 8, :w + 8]
i0 = np.stack([img[(b % 5):(b % 5) + h, (b % 3):(b % 3) + w] for b in range(B)]).astype(np.uint8)
i1 = np.stack([img[(b % 5) + 2:(b % 5) + 2 + h, (b % 3) + 3:(b % 3) + 3 + w] for b in range(B)]).astype(np.uint8)
fs = FlowStage(w, h, B)
fs.deepflow(i0, i1)
lib.sind_lab_ss_profile(None, 1)
t0 = time.perf_counter(); fs.deepflow(i0, i1); dt = time.perf_counter() - t0
out = (ctypes.c_ulonglong * 96)()
lib.sind_lab_ss_profile(out, 0)
print(f"B={B} {w}x{h}: {dt * 1e3:.1f} ms per batch")
print("wave : steps  busy/step  barrier/step | steps changing pairs  busy/step   (s_memtime ticks)")
for wv in range(16):
    b, wt, n, hb, hn = (out[wv * 6 + k] for k in range(5))
    if n:
        print(f"{wv:4d} : {n:9d} {b / n:9.1f} {wt / n:9.1f} | {hn:9d} {hb / max(hn, 1):9.1f}")
