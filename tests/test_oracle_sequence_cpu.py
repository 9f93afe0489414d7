"""CPU: the oracle's two ways through a sequence agree -- detect() frame by frame, and sequence_run() (dense flows of all frames first, on several threads,
then the stateful part in order): the long-sequence GPU parity tests use the second."""
import numpy as np

import oracle_lib as O
from sindslam_amd.synth import SyntheticStream, TUM3


def test_sequence_run_equals_detect_frame_by_frame():
    n = 5
    bgr, depth = SyntheticStream(seed=31).frames(0, n)
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    det = O.DynaDetect(bgr[0], bgr[0].copy(), *K)
    seq = O.sequence_run(bgr, depth, TUM3, threads=4)
    orb = O.ORBextractor(1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"])
    for f in range(1, n):
        dy, lb = det.detect(bgr[f], depth[f])
        assert np.array_equal(dy, seq["dyna"][f]) and np.array_equal(lb, seq["label"][f]), f
        k, d = orb.extract(O.bgr2gray(bgr[f], swap_rb=True), O.dilate15(dy))
        assert k.tobytes() == seq["keypoints"][f].tobytes() and np.array_equal(d, seq["descriptors"][f]), f
