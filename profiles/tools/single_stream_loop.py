"""One camera stream through the single-stream DynaDetect class, frame after frame (no oracle): the workload for a kernel trace of the
per-frame tail chain without contention.  usage: python3 profiles/tools/single_stream_loop.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sindslam_amd.dyna import DynaDetect
from sindslam_amd.synth import SyntheticStream, TUM3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bgr, depth = SyntheticStream(seed=777).frames(0, n)
gpu = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
for f in range(1, n):
    t0 = time.perf_counter(); gpu.DetectDynaArea(bgr[f], depth[f], f); print(f"frame {f}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
