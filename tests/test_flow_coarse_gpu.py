"""GPU parity of the one-launch chain over the one-workgroup pyramid levels (k_coarse_chain, flow_coarse.hip) against the CPU oracle and against the per-stage kernels,
bit for bit: every level size of the 384 x 288 and 768 x 432 pyramids that the chain takes, odd sizes at the limits of the two strip widths, and whole pyramids."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fs():
    from sindslam_amd.flow import FlowStage
    f = FlowStage(384, 288, max_batch=3)
    yield f
    f.close()


def _textured_pair(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.normal(0, 1, (h + 16, w + 16)).astype(np.float32)
    for _ in range(3):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) * np.float32(0.25)
    base = (base - base.min()) / (base.max() - base.min()) * np.float32(255)
    return np.rint(base[8:8 + h, 8:8 + w]).astype(np.float32), np.rint(base[6:6 + h, 11:11 + w]).astype(np.float32)


def chain_levels(w, h):
    return [(a, b) for a, b in O.deepflow_levels(w, h) if a * b <= 4096]


# every chain level of both configurations' pyramids (sizes repeat between them only by accident), plus shapes at the limits: 2-pixel strips up to 2048 pixels,
# 4-pixel strips beyond; odd widths (a last strip half outside the image), odd heights (one row parity has a row more), the narrowest / widest rows
LEVEL_SHAPES = sorted(set(chain_levels(384, 288) + chain_levels(768, 432) + [(64, 32), (63, 32), (65, 31), (45, 45), (91, 45), (90, 45), (120, 34), (26, 78), (8, 6), (5, 7), (124, 33)]))


@pytest.mark.parametrize("w,h", LEVEL_SHAPES)
def test_chain_level_equals_the_oracle(fs, w, h):
    i0, i1 = _textured_pair(w, h, 13 * w + h)
    rng = np.random.default_rng(w + 1000 * h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 5, 25, a, d, g, 1.6)
    # three images per launch, the middle one a different problem: a workgroup must not touch its neighbours' planes
    j0, j1 = _textured_pair(w, h, 17 * w + h + 1)
    pu, pv = O.varref(j0, j1, v0, u0, 5, 25, a, d, g, 1.6)
    I0 = np.stack([i0, j0, i0]); I1 = np.stack([i1, j1, i1]); U = np.stack([u0, v0, u0]); V = np.stack([v0, u0, v0])
    fs.set_coarse_chain(True)
    gu, gv = fs.varref_f32(I0, I1, U, V, 5, 25, a, d, g, 1.6)
    for b, (ru, rv) in enumerate([(ou, ov), (pu, pv), (ou, ov)]):
        assert np.array_equal(gu[b].view(np.uint32), ru.view(np.uint32)), (w, h, b, float(np.abs(gu[b] - ru).max()))
        assert np.array_equal(gv[b].view(np.uint32), rv.view(np.uint32)), (w, h, b, float(np.abs(gv[b] - rv).max()))
    try:
        fs.set_coarse_chain(False)
        su, sv = fs.varref_f32(I0, I1, U, V, 5, 25, a, d, g, 1.6)
    finally:
        fs.set_coarse_chain(True)
    assert np.array_equal(su.view(np.uint32), gu.view(np.uint32)) and np.array_equal(sv.view(np.uint32), gv.view(np.uint32))


def test_chain_with_refinement_defaults(fs):
    """VariationalRefinement's own defaults (alpha 20, delta 5, gamma 10, 5 x 5 iterations) on a chain-sized image"""
    w, h = 60, 44
    i0, i1 = _textured_pair(w, h, 5)
    rng = np.random.default_rng(9)
    u0 = rng.normal(0, 2.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 2.0, (h, w)).astype(np.float32)
    ou, ov = O.varref(i0, i1, u0, v0)
    gu, gv = fs.varref_f32(i0[None], i1[None], u0[None], v0[None])
    assert np.array_equal(gu[0].view(np.uint32), ou.view(np.uint32)) and np.array_equal(gv[0].view(np.uint32), ov.view(np.uint32))


def _gray_pair(frames, w, h):
    bgr, _ = frames
    return O.resize_u8(O.bgr2gray(bgr[2]), w, h), O.resize_u8(O.bgr2gray(bgr[0]), w, h)


@pytest.mark.parametrize("w,h", [(96, 72), (60, 50)])
def test_pyramid_wholly_inside_the_chain(frames, w, h):
    """a flow grid so small that EVERY level is a chain level: no up-sampling out of the chain, the result is taken from its planes"""
    from sindslam_amd.flow import FlowStage
    g0, g1 = _gray_pair(frames, w, h)
    f = FlowStage(w, h, max_batch=2)
    try:
        u, v = f.deepflow(np.stack([g0, g1]), np.stack([g1, g0]))
        of = O.deepflow(g0, g1); ob = O.deepflow(g1, g0)
        assert np.array_equal(u[0].view(np.uint32), of[..., 0].view(np.uint32)) and np.array_equal(v[0].view(np.uint32), of[..., 1].view(np.uint32))
        assert np.array_equal(u[1].view(np.uint32), ob[..., 0].view(np.uint32)) and np.array_equal(v[1].view(np.uint32), ob[..., 1].view(np.uint32))
    finally:
        f.close()


def test_deepflow_384x288_chain_equals_per_stage_and_oracle(fs, frames):
    """the 49-level pyramid: 16 chain levels in one launch + 33 per-stage levels == per-stage kernels on all 49 == the oracle; three pairs per launch"""
    g0, g1 = _gray_pair(frames, 384, 288)
    bgr, _ = frames
    g2 = O.resize_u8(O.bgr2gray(bgr[1]), 384, 288)
    i0 = np.stack([g0, g1, g0]); i1 = np.stack([g1, g0, g2])
    fs.set_coarse_chain(True); fs.set_latency_tiles(True)
    u, v = fs.deepflow(i0, i1)
    try:
        fs.set_coarse_chain(False); fs.set_latency_tiles(False); fs.set_level_up(False)
        su, sv = fs.deepflow(i0, i1)
    finally:
        fs.set_coarse_chain(True); fs.set_latency_tiles(True); fs.set_level_up(True)
    assert np.array_equal(u.view(np.uint32), su.view(np.uint32)) and np.array_equal(v.view(np.uint32), sv.view(np.uint32))
    o = O.deepflow(g0, g1)
    assert np.array_equal(u[0].view(np.uint32), o[..., 0].view(np.uint32)) and np.array_equal(v[0].view(np.uint32), o[..., 1].view(np.uint32))


def test_deepflow_768x432_chain_equals_per_stage():
    """the 57-level pyramid of the 1280 x 720 configuration"""
    from sindslam_amd.flow import FlowStage
    f = FlowStage(768, 432, max_batch=2)
    try:
        rng = np.random.default_rng(3)
        a = _textured_pair(768, 432, 21); b = _textured_pair(768, 432, 22)
        i0 = np.stack([a[0], b[0]]).astype(np.uint8); i1 = np.stack([a[1], b[1]]).astype(np.uint8)
        u, v = f.deepflow(i0, i1)
        f.set_coarse_chain(False); f.set_latency_tiles(False); f.set_level_up(False)
        su, sv = f.deepflow(i0, i1)
        assert np.array_equal(u.view(np.uint32), su.view(np.uint32)) and np.array_equal(v.view(np.uint32), sv.view(np.uint32))
    finally:
        f.close()


# k_sor_tile (1024-thread tiles, up to 13 iterations per launch): shapes whose plans are 13 + 12, 9 + 8 + 8, 7 + 6 + 6 + 6 and 5 x 5 iterations per launch at one to three
# images, tiles clipped at every image border, widths that are not multiples of the 4-pixel strips, odd heights
TILE_SHAPES = [(100, 80, 1), (104, 78, 3), (129, 67, 2), (153, 99, 1), (155, 70, 3), (300, 101, 1), (303, 65, 2), (384, 287, 1), (383, 288, 2), (384, 288, 3)]


@pytest.mark.parametrize("w,h,B", TILE_SHAPES)
def test_latency_tiles_equal_the_oracle(fs, w, h, B):
    i0, i1 = _textured_pair(w, h, 7 * w + h)
    rng = np.random.default_rng(w + 1000 * h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 5, 25, a, d, g, 1.6)
    fs.set_latency_tiles(True)
    gu, gv = fs.varref_f32(np.stack([i0] * B), np.stack([i1] * B), np.stack([u0] * B), np.stack([v0] * B), 5, 25, a, d, g, 1.6)
    for b in range(B):
        assert np.array_equal(gu[b].view(np.uint32), ou.view(np.uint32)), (w, h, b, float(np.abs(gu[b] - ou).max()))
        assert np.array_equal(gv[b].view(np.uint32), ov.view(np.uint32)), (w, h, b, float(np.abs(gv[b] - ov).max()))


def test_latency_tiles_refinement_defaults(fs, frames):
    """VariationalRefinement::create()->calc at 384 x 288 (5 x 5 iterations: one launch of five per fixed-point iteration) == the throughput kernels"""
    g0, g1 = _gray_pair(frames, 384, 288)
    rng = np.random.default_rng(4)
    u0 = rng.normal(0, 1.5, (1, 288, 384)).astype(np.float32); v0 = rng.normal(0, 1.5, (1, 288, 384)).astype(np.float32)
    fs.set_latency_tiles(True)
    u, v = fs.refine(g0[None], g1[None], u0, v0)
    try:
        fs.set_latency_tiles(False)
        su, sv = fs.refine(g0[None], g1[None], u0, v0)
    finally:
        fs.set_latency_tiles(True)
    assert np.array_equal(u.view(np.uint32), su.view(np.uint32)) and np.array_equal(v.view(np.uint32), sv.view(np.uint32))
