"""GPU parity: ORBextractor through the C ABI vs the CPU oracle — every integer stage bit-exact."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def grays(frames):
    bgr, _ = frames
    return [O.bgr2gray(bgr[i]) for i in range(3)]


@pytest.fixture(scope="module")
def orb():
    from sindslam_amd.orb import ORBextractor
    o = ORBextractor(1500, 1.2, 8, 15, 5)
    yield o
    o.close()


def test_tables(orb):
    t = orb.tables(); r = O.ORBextractor(1500, 1.2, 8, 15, 5).tables()
    for k in t:
        assert np.array_equal(t[k], r[k]), k
    assert t["per_level"].tolist() == [326, 271, 226, 189, 157, 131, 109, 91]


def test_stages_bitexact(orb, grays):
    ref = O.ORBextractor(1500, 1.2, 8, 15, 5)
    for g in grays[:2]:
        rk, rd = ref.extract(g)
        k, d = orb(g)
        for lv in range(8):
            assert np.array_equal(orb.image_pyramid(lv), ref.level_padded(lv)), f"pyramid level {lv}"
            a = orb.debug_fast(lv); b = ref.fast_keypoints(lv)
            assert len(a) == len(b), (lv, len(a), len(b))
            assert np.array_equal(a[:, 0], b["x"]) and np.array_equal(a[:, 1], b["y"]) and np.array_equal(a[:, 2], b["response"]), f"FAST level {lv}"
        sk, sd = orb.debug_selected()
        rs = np.concatenate([ref.selected(lv) for lv in range(8)])
        assert len(sk) == len(rs)
        for f in ("x", "y", "size", "response", "octave"):
            assert np.array_equal(sk[f], rs[f]), f
        assert np.array_equal(sk["angle"].view(np.uint32), rs["angle"].view(np.uint32)), np.abs(sk["angle"] - rs["angle"]).max()
        assert len(k) == len(rk) and k.tobytes() == rk.tobytes(), "final keypoints"
        assert np.array_equal(d, rd), "descriptors"


def test_mask_filter_and_fallback(orb, grays):
    g = grays[2]; h, w = g.shape
    ref = O.ORBextractor(1500, 1.2, 8, 15, 5)
    mask = np.zeros((h, w), np.uint8); mask[:, : w // 2] = 255; mask[100:200, 400:500] = 125
    rk, rd = ref.extract(g, mask); k, d = orb(g, mask)
    assert 250 <= len(k) < 1400 and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    full = np.full((h, w), 255, np.uint8)           # everything dynamic -> fewer than 250 survive -> all restored
    rk2, rd2 = ref.extract(g, full); k2, d2 = orb(g, full)
    rk0, _ = ref.extract(g)
    assert len(k2) == len(rk0) and k2.tobytes() == rk2.tobytes() and np.array_equal(d2, rd2)


def test_batch_and_other_params(grays):
    from sindslam_amd.orb import ORBextractor
    o = ORBextractor(1500, 1.2, 8, 20, 7)           # Bonn / D455 settings (reference Bonn.yaml:41,53,54, D455.yaml:41,53,54)
    ref = O.ORBextractor(1500, 1.2, 8, 20, 7)
    ks, ds = o.extract_batch(np.stack(grays))
    for g, k, d in zip(grays, ks, ds):
        rk, rd = ref.extract(g)
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    assert len(o(np.zeros((0, 0), np.uint8))[0]) == 0       # empty image: silent return
    flat = np.full((480, 640), 77, np.uint8)                # no corners anywhere
    k, d = o(flat); rk, rd = ref.extract(flat)
    assert len(k) == 0 and len(rk) == 0
    o.close()
