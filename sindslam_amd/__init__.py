"""sindslam_amd — MI355X-native DynaDetect + ORBextractor hot path.

Python mirror of the reference's C++ class API on top of the C ABI in include/sind_hip.h.
"""
import os as _os

# The pipeline drives ~45 HIP streams per handle; the runtime multiplexes them onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Six keep the flow slices'
# launches clear of the tail streams' (measured: profiles/r04/hw_queues.txt).  Only effective when set before the HIP runtime starts; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

from ._lib import SindError, SO_PATH  # noqa: F401,E402
