"""GPU: edge cases (SURVEY §8c asks for empty / degenerate inputs) and the 1280x720 configuration of BASELINE.json."""
import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd.synth import D455, TUM3, SyntheticStream

pytestmark = pytest.mark.gpu
K3 = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])


def test_no_valid_depth_and_static_scene(frames):
    """all-zero depth (everything invalid) and three identical frames (zero flow): no crash, legal output codes"""
    from sindslam_amd.dyna import DynaDetect
    bgr, depth = frames
    dd = DynaDetect(bgr[1], bgr[0], *K3)
    dy, lb = dd.DetectDynaArea(bgr[2], np.zeros_like(depth[2]), 2)
    assert not dy.any() and not lb.any()                      # no valid depth -> nothing static, nothing dynamic
    ref = O.DynaDetect(bgr[1], bgr[0], *K3); rd, rl = ref.detect(bgr[2], np.zeros_like(depth[2]))
    assert np.array_equal(dy, rd) and np.array_equal(lb, rl)
    dy, lb = dd.DetectDynaArea(bgr[3], depth[3], 3)           # recovers on the next good frame
    assert set(np.unique(dy)) <= {0, 125, 255} and (dy == 125).sum() > 10000
    st = DynaDetect(bgr[0], bgr[0], *K3)
    dy, lb = st.DetectDynaArea(bgr[0], depth[0], 2)           # identical frames: flow == 0 everywhere
    assert set(np.unique(dy)) <= {0, 125, 255}
    dd.close(); st.close()


def test_orb_textureless_and_tiny_feature_budget(frames):
    from sindslam_amd.orb import ORBextractor
    bgr, _ = frames; g = O.bgr2gray(bgr[2])
    o = ORBextractor(100, 1.2, 4, 20, 7); r = O.ORBextractor(100, 1.2, 4, 20, 7)
    k, d = o(g); rk, rd = r.extract(g)
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd) and 0 < len(k) <= 140
    noise = np.random.default_rng(0).integers(0, 256, (480, 640), dtype=np.uint8)      # maximum corner density
    k, d = o(noise); rk, rd = r.extract(noise)
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    o.close()


@pytest.mark.timeout(900)
def test_1280x720_d455():
    """BASELINE.json configs[4]: 1280x720, D455 intrinsics x2, depth factor 1000, FAST 20/7 (flow grid 768x432, 57 levels)"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    s = SyntheticStream(width=1280, height=720, intr=D455)
    bgr, depth = s.frames(0, 3)
    Kd = (s.fx, s.fy, s.cx, s.cy, D455["depth_factor"])
    gpu = DynaDetect(bgr[1], bgr[0], *Kd); ref = O.DynaDetect(bgr[1], bgr[0], *Kd)
    gd, gl = gpu.DetectDynaArea(bgr[2], depth[2], 2); rd, rl = ref.detect(bgr[2], depth[2])
    g = gpu.debug(); r = ref.debug()
    ff = np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]])
    assert np.array_equal(g["flow_full"].view(np.uint32), ff.view(np.uint32))
    assert np.array_equal(g["mask_high"], r["mask_high"]) and np.array_equal(g["occ1"], r["occ1"])
    u = np.logical_or(gd == 255, rd == 255).sum()
    assert u == 0 or np.logical_and(gd == 255, rd == 255).sum() / u >= 0.99
    gray = O.bgr2gray(bgr[2]); mask = gpu.dilate15(gd)           # D455.yaml:28 Camera.RGB: 0 -> BGR2GRAY for the tracker as well
    k, d = ORBextractor(1500, 1.2, 8, 20, 7)(gray, mask); rk, rdesc = O.ORBextractor(1500, 1.2, 8, 20, 7).extract(gray, mask)      # D455.yaml:41,53,54
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc)
    gpu.close()


@pytest.mark.timeout(900)
def test_1280x720_d455_three_level_flow_pyramid():
    """BASELINE.json configs[4] as written: "D455i 1280x720 ..., 3-level flow pyramid" -- the build-side option (DeepFlow's maxLayers made effective:
    only the finest three levels of the 0.95 pyramid, flow from zero at the third): flow bit-exact, masks / labels / ORB equal to the oracle
    run with the same option, over three frames so that the state roll is included"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    s = SyntheticStream(width=1280, height=720, intr=D455, motion_scale=0.5)       # three levels only resolve small motion
    bgr, depth = s.frames(0, 5)
    Kd = (s.fx, s.fy, s.cx, s.cy, D455["depth_factor"])
    gpu = DynaDetect(bgr[1], bgr[0], *Kd); ref = O.DynaDetect(bgr[1], bgr[0], *Kd)
    gpu.set_flow_max_levels(3); ref.set_flow_max_levels(3)
    orb = ORBextractor(1500, 1.2, 8, 20, 7); rorb = O.ORBextractor(1500, 1.2, 8, 20, 7)
    for t in range(2, 5):
        gd, gl = gpu.DetectDynaArea(bgr[t], depth[t], t); rd, rl = ref.detect(bgr[t], depth[t])
        g = gpu.debug(); r = ref.debug()
        ff = np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]])
        assert np.array_equal(g["flow_full"].view(np.uint32), ff.view(np.uint32)), t
        assert np.array_equal(gd, rd) and np.array_equal(gl, rl), t
        gray = O.bgr2gray(bgr[t]); mask = gpu.dilate15(gd)
        k, d = orb(gray, mask); rk, rdesc = rorb.extract(gray, mask)
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc), t
    gpu.close(); orb.close()


def test_bonn_configuration():
    """BASELINE.json configs[2]: Bonn intrinsics (Examples/RGB-D/Bonn.yaml), FAST 20/7, through DynaDetect -> dilate -> ORB for four frames"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    from sindslam_amd.synth import BONN
    s = SyntheticStream(seed=4242, intr=BONN, motion_scale=1.5)
    bgr, depth = s.frames(0, 6)
    Kb = (s.fx, s.fy, s.cx, s.cy, BONN["depth_factor"])
    gpu = DynaDetect(bgr[1], bgr[0], *Kb); ref = O.DynaDetect(bgr[1], bgr[0], *Kb)
    orb = ORBextractor(1500, 1.2, 8, BONN["ini_th"], BONN["min_th"]); rorb = O.ORBextractor(1500, 1.2, 8, BONN["ini_th"], BONN["min_th"])      # Bonn.yaml:41,53,54
    for t in range(2, 6):
        gd, gl = gpu.DetectDynaArea(bgr[t], depth[t], t); rd, rl = ref.detect(bgr[t], depth[t])
        u = np.logical_or(gd == 255, rd == 255).sum()
        assert u == 0 or np.logical_and(gd == 255, rd == 255).sum() / u >= 0.99          # IoU bar of BASELINE.json
        assert np.array_equal(gd, rd) and np.array_equal(gl, rl)
        gray = O.bgr2gray(bgr[t], swap_rb=True); mask = gpu.dilate15(gd)           # Bonn.yaml:28 Camera.RGB: 1 -> the tracker applies RGB2GRAY to the BGR buffer (src/Tracking.cc:246-251)
        k, d = orb(gray, mask); rk, rdesc = rorb.extract(gray, O.dilate15(rd))
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc)
    gpu.close(); orb.close()


@pytest.mark.timeout(900)
def test_sizes_the_region_grow_kernel_does_not_take_grow_on_the_host():
    """640 x 360 (the height is not a whole number of 16 x 16 blocks): the PEAC region grow kernel does not take the size (PeacGrowBatch::supports), so the frames are
    grown by the host statement of the same FIFO -- single-frame API and batched pipeline (B >= 4: the batched CalOccluded path), both equal to the oracle"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.pipeline import Pipeline
    s = SyntheticStream(width=640, height=360, seed=77)
    bgr, depth = s.frames(0, 4)
    Kd = (s.fx, s.fy, s.cx, s.cy, TUM3["depth_factor"])
    gpu = DynaDetect(bgr[1], bgr[0], *Kd); ref = O.DynaDetect(bgr[1], bgr[0], *Kd)
    outs = []
    for f in (2, 3):
        gd, gl = gpu.DetectDynaArea(bgr[f], depth[f], f); rd, rl = ref.detect(bgr[f], depth[f])
        assert np.array_equal(gd, rd) and np.array_equal(gl, rl), f
        outs.append((gd, gl))
    # ORB at this size: pyramid level 6 is 214 x 121 -> two rows of 45-px FAST cells (+ 6 px overlap = 51; the window limit used to be 48)
    from sindslam_amd.orb import ORBextractor
    gray = O.bgr2gray(bgr[3], swap_rb=True); mask = gpu.dilate15(outs[1][0])
    k, d = ORBextractor(1500, 1.2, 8, 15, 5)(gray, mask); rk, rdesc = O.ORBextractor(1500, 1.2, 8, 15, 5).extract(gray, mask)
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc) and len(k) > 300
    gpu.close()
    pipe = Pipeline(4, 2, 640, 360, *Kd, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1)
    for k in range(4):
        pipe.prime(k, bgr[1], bgr[0])
    pipe.process(np.stack([bgr[2:4]] * 4), np.stack([depth[2:4]] * 4))
    for k in range(4):
        for t in range(2):
            assert np.array_equal(pipe.dyna[k, t], outs[t][0]) and np.array_equal(pipe.label[k, t], outs[t][1]), (k, t)
    kk, dd = pipe.keypoints(0, 1); assert kk.tobytes() == rk.tobytes() and np.array_equal(dd, rdesc)
    assert pipe.grow_share() >= 0
    pipe.close()
