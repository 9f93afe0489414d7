import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench
from sindslam_amd import sequence as SQ
import sindslam_amd.synth as SY
# wrap Pipeline.replay / get_state / set_state with timers
from sindslam_amd.pipeline import Pipeline
T = {"replay": [], "get_state": 0.0, "set_state": 0.0, "hashes": 0.0}
_r = Pipeline.replay
def replay(self, tag, first, last):
    t0 = time.perf_counter(); _r(self, tag, first, last); T["replay"].append((round((time.perf_counter() - t0) * 1e3, 1), int((np.asarray(last) > np.asarray(first)).sum()), int((np.asarray(last) - np.asarray(first)).max())))
Pipeline.replay = replay
for name in ("get_state", "set_state"):
    f = getattr(Pipeline, name)
    def mk(f, name):
        def w(self, *a, **k):
            t0 = time.perf_counter(); r = f(self, *a, **k); T[name] += time.perf_counter() - t0; return r
        return w
    setattr(Pipeline, name, mk(f, name))
sys.argv = ["bench.py", "--workload", "sequence", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-exact-leg"] + sys.argv[1:]
import io, contextlib
bench.main()
print("replay calls (ms, live streams, frames):", T["replay"], "get_state %.1f ms set_state %.1f ms" % (T["get_state"] * 1e3, T["set_state"] * 1e3), file=sys.stderr)
