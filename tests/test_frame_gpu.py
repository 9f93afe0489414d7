"""GPU: Frame post-ORB steps (sind_frame_post_orb, reference src/Frame.cc:143-170) against the oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_golden import FRAME_CALIBS  # noqa: E402

KEYS = ("keys_un", "u_right", "depth", "cell", "grid_start", "grid_idx")


def _stage(cal, B, cap=4096, w=640, h=480):
    from sindslam_amd.frame import FramePostORB
    return FramePostORB(w, h, cal[0], cal[1], cal[2], cal[3], cal[9], cal[10], dist=cal[4:9], max_batch=B, cap=cap)


def _same(got, want):
    for key in KEYS:
        assert getattr(got, key).tobytes() == want[key].tobytes(), key


@pytest.mark.parametrize("name", ["tum3", "tum1"])
def test_matches_fixture_and_oracle_batched(name, stream):
    import oracle_lib as O
    from sindslam_amd.orb import ORBextractor
    cal = FRAME_CALIBS[name]; g = np.load(os.path.join(GOLD, "frame_post.npz"))
    bgr, depth = stream.frames(2, 3)
    orb = ORBextractor(1500, 1.2, 8, 15, 5)
    kps, _ = orb.extract_batch(np.stack([O.bgr2gray(b) for b in bgr]))
    st = _stage(cal, 3)
    out = st(kps, depth)
    _same(out[0], {k: g[f"{name}_{k}"] for k in KEYS})                      # frame 2 is the fixture's frame
    assert st.bounds.tobytes() == g[f"{name}_bounds"].tobytes()
    for b in range(3):
        _same(out[b], O.frame_post_orb(cal, kps[b]["x"], kps[b]["y"], depth[b]))
    assert np.array_equal(out[1].grid(5, 7), np.nonzero(out[1].cell == 5 * 48 + 7)[0])
    st.close(); orb.close()


def test_device_depth_pointer_and_dense_random_keypoints():
    import torch
    import oracle_lib as O
    from sindslam_amd.orb import KP_DTYPE
    rng = np.random.default_rng(7); cal = FRAME_CALIBS["tum1"]; n = 20000
    k = np.zeros(n, KP_DTYPE); k["x"] = rng.uniform(0, 639.99, n).astype(np.float32); k["y"] = rng.uniform(0, 479.99, n).astype(np.float32)
    depth = rng.integers(0, 30000, (1, 480, 640)).astype(np.uint16); depth[0, ::3] = 0          # a third of the rows without depth
    st = _stage(cal, 1, cap=n)
    dd = torch.from_numpy(depth.view(np.int16)).cuda()
    out = st([k], int(dd.data_ptr()))[0]
    _same(out, O.frame_post_orb(cal, k["x"], k["y"], depth[0]))
    assert (out.depth < 0).any() and (out.cell < 0).any()                   # both "no depth" and "outside the undistorted bounds" occur
    st.close()


def test_empty_frame_and_capacity_error():
    from sindslam_amd import SindError
    from sindslam_amd.orb import KP_DTYPE
    st = _stage(FRAME_CALIBS["tum3"], 2, cap=16)
    depth = np.full((2, 480, 640), 5000, np.uint16)
    one = np.zeros(1, KP_DTYPE); one["x"] = 100.5; one["y"] = 50.25
    out = st([np.zeros(0, KP_DTYPE), one], depth)
    assert len(out[0].keys_un) == 0 and not out[0].grid_start.any()
    assert out[1].depth[0] == np.float32(5000) * np.float32(1.0 / 5000.0) and out[1].cell[0] == 10 * 48 + 5
    with pytest.raises(SindError):
        st([np.zeros(17, KP_DTYPE), one], depth)
    bad = one.copy(); bad["x"] = 640.0
    with pytest.raises(SindError):
        st([bad, one], depth)
    st.close()
