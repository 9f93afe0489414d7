"""GPU: sind_cloud_generate (generatePointCloud of reference octomap_pub/src/pubPointCloud.cc:471-668) against the oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert np.array_equal(got["occlusion"], want["occlusion"]) and np.array_equal(got["label_count"], want["label_count"]) and np.array_equal(got["kept"], want["kept"])
    assert got["points"].tobytes() == want["points"].tobytes()


def test_stream_keyframes_batched(stream):
    import oracle_lib as O
    import cloud_scene as S
    from sindslam_amd.cloud import CloudGenerator
    scenes = [S.keyframe_pair(stream, t, gap) for t, gap in ((8, 2), (12, 1))]
    cam5 = scenes[0][0]
    gen = CloudGenerator(*cam5, max_batch=2)
    keys = ("imgRGB", "imgDepth", "imgDepthLast", "imgDynaMask", "imgDynaMaskLast", "imgLabel", "poseRelative", "Twc")
    out = gen.generatePointCloud(*[np.stack([s[1][k] for s in scenes]) for k in keys])
    for got, (_, a) in zip(out, scenes):
        want = O.generate_point_cloud(cam5, *[a[k] for k in keys])
        _same(got, want)
        assert len(got["points"]) > 30000 and np.isnan(got["points"]["x"]).any() and want["occlusion"].sum() > 0
    gen.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_random_labels_rejected_clusters_and_odd_size(seed):
    import oracle_lib as O
    from sindslam_amd.cloud import CloudGenerator
    rng = np.random.default_rng(seed); h, w = 123, 201                                    # odd sizes: ragged last row / column of the stride-2 grid
    depth = rng.integers(0, 60000, (h, w)).astype(np.uint16); last = (depth.astype(np.int32) + rng.integers(-3000, 3000, (h, w))).clip(0, 65535).astype(np.uint16)
    lab = np.kron(rng.integers(0, 14, (h // 8 + 1, w // 8 + 1)), np.ones((8, 8), np.int64))[:h, :w].astype(np.uint8)      # labels 12, 13 are skipped (:562)
    dyna = (rng.random((h, w)) < 0.2).astype(np.uint8) * 255; dl = (rng.random((h, w)) < 0.1).astype(np.uint8) * 255
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    a = np.deg2rad(2.0); rel = np.eye(4); rel[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]; rel[:3, 3] = (0.03, -0.01, 0.02)
    twc = np.linalg.inv(rel) @ np.diag([1, 1, 1, 1.0]); twc[:3, 3] += (1.0, 2.0, 3.0)
    cam5 = [180.0, 181.0, 100.3, 61.7, 5000.0]
    gen = CloudGenerator(*cam5, width=w, height=h)
    got, = gen.generatePointCloud(bgr[None], depth[None], last[None], dyna[None], dl[None], lab[None], rel[None], twc[None])
    want = O.generate_point_cloud(cam5, bgr, depth, last, dyna, dl, lab, rel, twc)
    _same(got, want)
    assert 0 < want["kept"].sum() < 12                                                    # both branches of the rejection rule
    gen.close()


def test_all_rejected_but_cluster_zero_and_capacity():
    from sindslam_amd import SindError
    from sindslam_amd._lib import check, lib, ptr
    from sindslam_amd.cloud import CloudGenerator, POINT_DTYPE
    h, w = 48, 64
    depth = np.full((1, h, w), 10000, np.uint16); bgr = np.zeros((1, h, w, 3), np.uint8); lab = np.ones((1, h, w), np.uint8); lab[0, :, :8] = 0
    last = np.full((1, h, w), 255, np.uint8); z = np.zeros((1, h, w), np.uint8)
    gen = CloudGenerator(50.0, 50.0, 32.0, 24.0, 5000.0, width=w, height=h)
    got, = gen.generatePointCloud(bgr, depth, depth, z, last, lab, np.eye(4)[None], np.eye(4)[None])
    assert got["kept"].tolist() == [1, 0] + [1] * 10 and len(got["points"]) == (h // 2) * 4
    pts = np.zeros(10, POINT_DTYPE); n = np.zeros(1, np.int32); I = np.ascontiguousarray(np.eye(4)[None])
    with pytest.raises(SindError):
        check(lib().sind_cloud_generate(gen._h, 1, ptr(bgr), ptr(depth), ptr(depth), ptr(z), ptr(last), ptr(lab), ptr(I), ptr(I), 0, ptr(pts), 10, ptr(n), None, None, None))
    gen.close()
