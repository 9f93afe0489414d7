import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
n = int(sys.argv[1]); rest = sys.argv[2:]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from sindslam_amd._lib import lib
lib().sind_debug_set_kmeans_fused_max(n)
import bench
sys.argv = ["bench.py"] + rest
bench.main()
