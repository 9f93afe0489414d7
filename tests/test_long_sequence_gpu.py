"""GPU: a longer run (state carried over 14 frames, large-motion frames included) against the oracle: the metric's
'mask IoU vs CPU ref' (mean / min), label agreement and ORB equality per frame."""
import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd.synth import SyntheticStream, TUM3

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
@pytest.mark.parametrize("motion,nframes,expect_large_motion", [(1.6, 16, False), (6.0, 8, True)])
def test_stateful_sequences(motion, nframes, expect_large_motion):
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    s = SyntheticStream(seed=777, motion_scale=motion)       # motion 6.0 drives the second DeepFlow pass (DD:1121-1131)
    bgr, depth = s.frames(0, nframes)
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    ref = O.DynaDetect(bgr[1], bgr[0], *K); gpu = DynaDetect(bgr[1], bgr[0], *K)
    orb = ORBextractor(1500, 1.2, 8, 15, 5); orbr = O.ORBextractor(1500, 1.2, 8, 15, 5)
    ious, lab_mis, lm, kp_equal = [], [], 0, 0
    n = nframes - 2
    for t in range(2, nframes):
        rd, rl = ref.detect(bgr[t], depth[t]); gd, gl = gpu.DetectDynaArea(bgr[t], depth[t], t)
        r = ref.debug(); g = gpu.debug()
        assert g["info"][0] == r["info"][0]; lm += int(g["info"][0])
        assert np.array_equal(g["flow_full"].view(np.uint32), np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]]).view(np.uint32)), t
        u = np.logical_or(gd == 255, rd == 255).sum()
        ious.append(1.0 if u == 0 else np.logical_and(gd == 255, rd == 255).sum() / u); lab_mis.append((gl != rl).mean())
        gray = O.bgr2gray(bgr[t], swap_rb=True)
        k, d = orb(gray, gpu.dilate15(gd)); rk, rdesc = orbr.extract(gray, O.dilate15(rd))
        kp_equal += int(k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc))
    print(f"IoU mean {np.mean(ious):.4f} min {np.min(ious):.4f}; label mismatch max {np.max(lab_mis):.5f}; large-motion frames {lm}; ORB-equal frames {kp_equal}/{n}")
    assert np.min(ious) >= 0.99 and np.max(lab_mis) <= 0.01
    assert kp_equal >= n - 1
    assert (lm > 0) == expect_large_motion
    gpu.close(); orb.close()
