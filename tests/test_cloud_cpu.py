"""CPU: oracle restatement of the mapping consumer's generatePointCloud (oracle/cloud.hpp) against domain properties."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
import cloud_scene as S  # noqa: E402


def test_cloud_structure_and_cluster_rule():
    from sindslam_amd.synth import SyntheticStream
    st = SyntheticStream(seed=12345)
    cam5, a = S.keyframe_pair(st, 8)
    r = O.generate_point_cloud(cam5, a["imgRGB"], a["imgDepth"], a["imgDepthLast"], a["imgDynaMask"], a["imgDynaMaskLast"], a["imgLabel"], a["poseRelative"], a["Twc"])
    lab = a["imgLabel"]; sub = lab[::2, ::2]
    assert np.array_equal(r["label_count"], np.bincount(lab.ravel(), minlength=12)[:12])
    kept = (np.arange(12) == 0) | (r["occlusion"] * 9 <= 0.4 * r["label_count"])
    assert np.array_equal(r["kept"].astype(bool), kept)
    assert len(r["points"]) == sum(int((sub == i).sum()) for i in range(12) if kept[i])
    # order: cluster by cluster, raster order inside -> colours of the first cluster's points follow the image
    first = a["imgRGB"][::2, ::2][sub == 0]
    p = r["points"][:len(first)]
    assert np.array_equal(np.stack([p["b"], p["g"], p["r"]], 1), first)
    # finite points are the back-projection moved by Twc; masked pixels are NaN
    m0 = (a["imgDynaMask"][::2, ::2] >= 240)[sub == 0]
    assert np.isnan(p["x"][m0]).all()
    d = (a["imgDepth"][::2, ::2][sub == 0].astype(np.float64) / cam5[4])
    ok = ~m0 & (d >= 0.01) & (d <= 10)
    vv, uu = np.nonzero(sub == 0); uu, vv = uu * 2, vv * 2
    Xc = np.stack([(uu - cam5[2]) * d / cam5[0], (vv - cam5[3]) * d / cam5[1], d], 1)
    Xw = Xc @ a["Twc"][:3, :3].T + a["Twc"][:3, 3]
    assert np.allclose(np.stack([p["x"], p["y"], p["z"]], 1)[ok], Xw[ok], atol=1e-4)
    assert np.isfinite(p["x"][ok]).all() and (p["a"] == 255).all()
    # the walkers moved between the key frames: some cluster collects occlusion votes
    assert r["occlusion"].sum() > 0


def test_identity_pose_static_depth_has_no_votes_outside_last_mask():
    h, w = 48, 64
    depth = np.full((h, w), 10000, np.uint16); bgr = np.zeros((h, w, 3), np.uint8); lab = np.ones((h, w), np.uint8); lab[:, w // 2:] = 3
    z = np.zeros((h, w), np.uint8); last = z.copy(); last[:, w // 2:] = 255
    r = O.generate_point_cloud([50.0, 50.0, 32.0, 24.0, 5000.0], bgr, depth, depth, z, last, lab, np.eye(4), np.eye(4))
    assert r["occlusion"][1] == 0 and r["occlusion"][3] == (h // 2) * (w // 4)           # isDynaLast votes only
    assert r["kept"][1] == 1 and r["kept"][3] == 0 and len(r["points"]) == (h // 2) * (w // 4)
