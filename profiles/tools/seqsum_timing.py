"""k_km_seqsum duration by run length and data shape, from a kernel trace of this script:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/tools/seqsum_timing.py ; python3 profiles/tools/seqsum_timing.py --report <dir>"""
import csv, ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rng = np.random.default_rng(5)
CASES = [("zeros 100k", np.zeros(100_000, np.float32)), ("ones 1k", np.ones(1000, np.float32)), ("ones 100k", np.ones(100_000, np.float32)),
         ("uniform(0.75, 9) 400", rng.uniform(0.75, 9, 400).astype(np.float32)), ("uniform(0.75, 9) 25k", rng.uniform(0.75, 9, 25_000).astype(np.float32)),
         ("uniform(0.75, 9) 100k", rng.uniform(0.75, 9, 100_000).astype(np.float32)), ("normal(0, 1.5) 100k", rng.normal(0, 1.5, 100_000).astype(np.float32)),
         ("row sign flips 100k", np.tile(np.concatenate([-rng.uniform(0, 2, 300), rng.uniform(0, 2, 300)]).astype(np.float32), 167))]
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_km_seqsum" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for (name, x), r in zip(CASES * 3, rows):
        print(f"{name:28s} n {x.size:7d}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
else:
    from sindslam_amd._lib import check, lib, ptr
    out = C.c_float(0)
    for rep in range(3):
        for name, x in CASES:
            check(lib().sind_debug_seqsum(ptr(x), int(x.size), 0, C.byref(out)))
