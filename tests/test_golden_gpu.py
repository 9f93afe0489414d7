"""GPU: the HIP path against the committed golden fixtures (tests/golden/, produced by the oracle)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_flow_fixture():
    from sindslam_amd.flow import FlowStage
    g = np.load(os.path.join(GOLD, "flow_96x72.npz"))
    f = FlowStage(96, 72, max_batch=1)
    u, v = f.deepflow(g["i0"][None], g["i1"][None])
    assert np.array_equal(u[0].view(np.uint32), g["deep"][..., 0].view(np.uint32)) and np.array_equal(v[0].view(np.uint32), g["deep"][..., 1].view(np.uint32))
    ru, rv = f.refine(g["i0"][None], g["i1"][None], -u, -v)
    assert np.array_equal(np.stack([ru[0], rv[0]]).view(np.uint32), g["refined"].view(np.uint32))
    f.close()


def test_orb_fixture(stream):
    import oracle_lib as O
    from sindslam_amd.orb import ORBextractor
    g = np.load(os.path.join(GOLD, "orb_frame2.npz"))
    bgr, _ = stream.frames(2, 1); gray = O.bgr2gray(bgr[0])
    orb = ORBextractor(1500, 1.2, 8, 15, 5)
    k, d = orb(gray)
    assert k.tobytes() == g["kps"].tobytes() and np.array_equal(d, g["desc"])
    mask = np.zeros((480, 640), np.uint8); mask[:, :320] = 255
    k, d = orb(gray, mask)
    assert k.tobytes() == g["kps_masked"].tobytes() and np.array_equal(d, g["desc_masked"])
    orb.close()


def test_dyna_fixture(frames):
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.synth import TUM3
    g = np.load(os.path.join(GOLD, "dyna_frames23.npz"))
    bgr, depth = frames
    dd = DynaDetect(bgr[1], bgr[0], TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    for t in (2, 3):
        dy, lb = dd.DetectDynaArea(bgr[t], depth[t], t); dbg = dd.debug()
        ref = np.unpackbits(g[f"dyna255_{t}"])[: 480 * 640].reshape(480, 640).astype(bool)
        u = np.logical_or(dy == 255, ref).sum(); iou = 1.0 if u == 0 else np.logical_and(dy == 255, ref).sum() / u
        assert iou >= 0.99, (t, iou)                                             # north_star tolerance
        assert np.array_equal(dbg["H"], g[f"H_{t}"]) and np.array_equal(dbg["thr"], g[f"thr_{t}"]) and np.array_equal(dbg["hist"], g[f"hist_{t}"])
        assert np.array_equal(np.packbits(dbg["occ1"] > 0), g[f"occ1_{t}"]) and np.array_equal(np.packbits(dbg["occ2"] > 0), g[f"occ2_{t}"])
    dd.close()
