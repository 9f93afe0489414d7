"""sindslam_amd — MI355X-native DynaDetect + ORBextractor hot path.

Python mirror of the reference's C++ class API on top of the C ABI in include/sind_hip.h.
"""
from ._lib import SindError, SO_PATH  # noqa: F401
