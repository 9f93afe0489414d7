"""CPU: oracle restatement of the Frame post-ORB steps (oracle/frame.hpp) against its fixture and against domain properties."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_golden import FRAME_CALIBS  # noqa: E402


def _inputs():
    from sindslam_amd.synth import SyntheticStream
    kps = np.load(os.path.join(GOLD, "orb_frame2.npz"))["kps"]
    depth = SyntheticStream(seed=12345).frames(2, 1)[1][0]
    return kps, depth


def test_oracle_matches_fixture():
    kps, depth = _inputs(); g = np.load(os.path.join(GOLD, "frame_post.npz"))
    for name, cal in FRAME_CALIBS.items():
        r = O.frame_post_orb(cal, kps["x"], kps["y"], depth)
        for key, val in r.items():
            assert val.tobytes() == g[f"{name}_{key}"].tobytes(), (name, key)


def test_undistort_is_identity_without_distortion_and_inverts_the_distortion_model():
    kps, depth = _inputs()
    r = O.frame_post_orb(FRAME_CALIBS["tum3"], kps["x"], kps["y"], depth)
    assert np.array_equal(r["keys_un"][:, 0], kps["x"]) and np.array_equal(r["keys_un"][:, 1], kps["y"])       # mvKeysUn = mvKeys (Frame.cc:479-483)
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(np.float32(v)) for v in FRAME_CALIBS["tum1"][:9]]
    un = O.frame_post_orb(FRAME_CALIBS["tum1"], kps["x"], kps["y"], depth)["keys_un"].astype(np.float64)
    x, y = (un[:, 0] - cx) / fx, (un[:, 1] - cy) / fy; r2 = x * x + y * y; rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd, yd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    assert np.abs(xd * fx + cx - kps["x"]).max() < 5e-3 and np.abs(yd * fy + cy - kps["y"]).max() < 5e-3      # 5 fixed-point iterations


def test_stereo_from_rgbd_and_grid_properties():
    kps, depth = _inputs(); cal = FRAME_CALIBS["tum1"]
    r = O.frame_post_orb(cal, kps["x"], kps["y"], depth)
    raw = depth[kps["y"].astype(np.int32), kps["x"].astype(np.int32)]
    d = raw.astype(np.float32) * np.float32(cal[10])
    assert np.array_equal(r["depth"], np.where(raw > 0, d, np.float32(-1)))
    ur = r["keys_un"][:, 0] - np.float32(cal[9]) / np.where(raw > 0, d, np.float32(1))
    assert np.array_equal(r["u_right"], np.where(raw > 0, ur, np.float32(-1)).astype(np.float32))
    # mGrid: every keypoint with a valid cell appears exactly once, in ascending order inside its cell
    gs, gi, cell = r["grid_start"], r["grid_idx"], r["cell"]
    assert gs[0] == 0 and gs[-1] == len(gi) == (cell >= 0).sum() and np.all(np.diff(gs) >= 0)
    for c in np.unique(cell[cell >= 0]):
        assert np.array_equal(gi[gs[c]:gs[c + 1]], np.nonzero(cell == c)[0])
    b = r["bounds"]; wInv, hInv = np.float32(64) / (b[1] - b[0]), np.float32(48) / (b[3] - b[2])
    px = np.floor((r["keys_un"][:, 0] - b[0]) * wInv + np.float32(0.5)).astype(np.int32)      # round() of non-negative values
    ok = cell >= 0
    assert np.array_equal(cell[ok] // 48, px[ok])


def test_no_keypoints():
    depth = np.zeros((48, 64), np.uint16)
    r = O.frame_post_orb(FRAME_CALIBS["tum3"], np.zeros(0, np.float32), np.zeros(0, np.float32), depth)
    assert len(r["grid_idx"]) == 0 and not r["grid_start"].any()
