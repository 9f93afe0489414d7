"""k_peac_grow alone: BFS levels, seeds and (under rocprofv3 --kernel-trace) the kernel's duration for ONE frame per launch at 640 x 480 and 1280 x 720.
usage: python3 profiles/tools/grow_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sindslam_amd._lib import check, lib, ptr
from sindslam_amd.synth import D455, TUM3, SyntheticStream
PP = 127 * 127
for name, w, h, intr in (("640x480", 640, 480, TUM3), ("1280x720", 1280, 720, dict(D455, fx=D455["fx"] * 2, fy=D455["fy"] * 2, cx=D455["cx"] * 2, cy=D455["cy"] * 2))):
    _, depth = SyntheticStream(w, h, seed=12345, intr=intr).frames(0, 6)
    for k in range(6):
        d = np.ascontiguousarray(depth[k:k + 1], np.uint16)
        mg = np.zeros((1, h * w), np.int8); mh = np.zeros((1, h * w), np.int8); pg = np.zeros((1, PP), np.uint8); ph = np.zeros((1, PP), np.uint8); st = np.zeros((1, 4), np.int32)
        t0 = time.perf_counter()
        check(lib().sind_debug_peac_grow(ptr(d), 1, w, h, C.c_float(intr["fx"]), C.c_float(intr["fy"]), C.c_float(intr["cx"]), C.c_float(intr["cy"]), C.c_float(intr["depth_factor"]), 0, ptr(mg), ptr(mh), ptr(pg), ptr(ph), ptr(st)))
        print(f"{name} frame {k}: status {st[0, 0]} levels {st[0, 1]} seeds {st[0, 2]} planes {st[0, 3]}  seeds/level {st[0, 2] / max(st[0, 1], 1):.0f}  (call incl. host FIFO {1e3 * (time.perf_counter() - t0):.1f} ms)", flush=True)
