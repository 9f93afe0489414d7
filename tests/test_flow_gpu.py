"""GPU parity: dense-flow stage (libsind_hip via the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fs():
    from sindslam_amd.flow import FlowStage
    f = FlowStage(384, 288, max_batch=4)
    yield f
    f.close()


def _small_pair(frames, w, h):
    bgr, _ = frames
    g0 = O.resize_u8(O.bgr2gray(bgr[2]), w, h); g1 = O.resize_u8(O.bgr2gray(bgr[0]), w, h)
    return g0, g1


def test_levels_match_oracle(fs):
    assert fs.levels() == O.deepflow_levels(384, 288)
    assert len(fs.levels()) == 49 and fs.levels()[-1] == (33, 26)


@pytest.mark.parametrize("w,h", [(33, 26), (64, 48), (97, 61), (128, 96)])
def test_varref_one_level_bitexact(fs, frames, w, h):
    g0, g1 = _small_pair(frames, w, h)
    rng = np.random.default_rng(w * 1000 + h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    i0 = g0.astype(np.float32); i1 = g1.astype(np.float32)
    # DeepFlow-level parameters (4*alpha, delta/3, gamma/3, 5 x 25) -- float32 arithmetic as in deepflow.cpp
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 5, 25, a, d, g, 1.6)
    gu, gv = fs.varref_f32(np.stack([i0, i0]), np.stack([i1, i1]), np.stack([u0, u0]), np.stack([v0, v0]), 5, 25, a, d, g, 1.6)
    for b in range(2):     # both batch entries identical and equal to the oracle
        assert np.array_equal(gu[b].view(np.uint32), ou.view(np.uint32)), np.abs(gu[b] - ou).max()
        assert np.array_equal(gv[b].view(np.uint32), ov.view(np.uint32)), np.abs(gv[b] - ov).max()


def test_deepflow_small_bitexact(frames):
    from sindslam_amd.flow import FlowStage
    w, h = 96, 72
    g0, g1 = _small_pair(frames, w, h)
    f = FlowStage(w, h, max_batch=2)
    assert f.levels() == O.deepflow_levels(w, h)
    u, v = f.deepflow(np.stack([g0, g1]), np.stack([g1, g0]))
    of = O.deepflow(g0, g1); ob = O.deepflow(g1, g0)
    assert np.array_equal(u[0].view(np.uint32), of[..., 0].view(np.uint32)), np.abs(u[0] - of[..., 0]).max()
    assert np.array_equal(v[0].view(np.uint32), of[..., 1].view(np.uint32))
    assert np.array_equal(u[1].view(np.uint32), ob[..., 0].view(np.uint32))
    assert np.array_equal(v[1].view(np.uint32), ob[..., 1].view(np.uint32))
    f.close()


def test_deepflow_full_size(fs, frames):
    """384x288 (the reference's flow grid): tolerance stated by SURVEY §8c is 1e-3 px; we expect bit equality."""
    g0, g1 = _small_pair(frames, 384, 288)
    u, v = fs.deepflow(g0[None], g1[None])
    o = O.deepflow(g0, g1)
    err = max(np.abs(u[0] - o[..., 0]).max(), np.abs(v[0] - o[..., 1]).max())
    assert err <= 1e-3, err
    assert np.array_equal(u[0].view(np.uint32), o[..., 0].view(np.uint32)) and np.array_equal(v[0].view(np.uint32), o[..., 1].view(np.uint32))
    # refinement on top of the negated flow (reference DynaDetect.cc:1080, 1133-1143)
    ru, rv = fs.refine(g0[None], g1[None], -u, -v)
    ou, ov = O.varref(g0.astype(np.float32), g1.astype(np.float32), -o[..., 0], -o[..., 1])
    assert np.array_equal(ru[0].view(np.uint32), ou.view(np.uint32)) and np.array_equal(rv[0].view(np.uint32), ov.view(np.uint32))


def test_sor_variants_agree_bitwise(fs, frames):
    """both fused register-resident SOR kernels (IEEE division / reciprocal division; several fuse depths and tile widths)
    == one-launch-per-colour SOR, bit for bit"""
    g0, g1 = _small_pair(frames, 384, 288)
    i0 = np.stack([g0, g1]); i1 = np.stack([g1, g0])
    try:
        from sindslam_amd._lib import lib
        lab = bool(lib().sind_lab_build())           # IEEE-division / reciprocal-plane / 1x4-strip variants and fuse plans exist in lab builds only
        fs.set_sor_variant(0, 5, 64, 64); ru, rv = fs.deepflow(i0, i1)
        for mode, fuse, tw, th in [(2, 5, 64, 64), (2, 3, 64, 64), (2, 7, 64, 64), (2, 1, 64, 64), (1, 5, 64, 64), (1, 3, 64, 64), (1, 7, 64, 64), (1, 5, 128, 64),
                                   (1, 1, 64, 64), (5, 5, 64, 64), (4, 0, 64, 64), (4, 5, 64, 64), (4, 3, 64, 64), (4, 5, 128, 64), (1, 0, 64, 64), (3, 3, 64, 48), (3, 5, 64, 48), (3, 1, 64, 48), (3, 2, 64, 32), (3, 4, 96, 64), (3, 5, 128, 48), (3, 3, 48, 64)]:
            if not lab and (mode not in (0, 4, 5) or fuse == 0):
                continue
            fs.set_sor_variant(mode, fuse, tw, th); u, v = fs.deepflow(i0, i1)
            assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32)), (mode, fuse, tw, th)
    finally:
        fs.set_sor_variant()


@pytest.fixture(scope="module")
def fs_big():
    from sindslam_amd.flow import FlowStage
    f = FlowStage(768, 432, max_batch=2)
    yield f
    f.close()


def _textured_pair(w, h, seed):
    """a band-limited random texture and a copy shifted / warped by a few pixels (u8 values as float32, like DeepFlow's level images)"""
    rng = np.random.default_rng(seed)
    base = rng.normal(0, 1, (h + 16, w + 16)).astype(np.float32)
    for _ in range(3):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) * np.float32(0.25)
    base = (base - base.min()) / (base.max() - base.min()) * np.float32(255)
    i0 = np.rint(base[8:8 + h, 8:8 + w]).astype(np.float32); i1 = np.rint(base[6:6 + h, 11:11 + w]).astype(np.float32)
    return i0, i1


# widths that cut into 1, 2, 3 and 6 column strips of the streaming kernel (a strip holds at most 152 columns), last strips whose width is not a multiple of
# four (w % 4 = 1, 2, 3), odd and even heights; every size is beyond the one-workgroup kernel, so mode 5 runs k_sor_stream on it
STREAM_SHAPES = [(129, 67), (152, 65), (153, 99), (155, 70), (300, 101), (302, 73), (303, 65), (384, 287), (385, 288), (768, 431), (767, 432), (612, 99)]


@pytest.mark.parametrize("w,h", STREAM_SHAPES)
def test_streaming_solver_equals_the_oracle_on_strip_cuts(fs_big, w, h):
    """the kernel the bench runs (k_sor_stream, solver mode 5 = streaming on every level it fits) against the ORACLE's VariationalRefinement directly, at widths
    that exercise every column-strip cut and heights that are odd: 5 fixed-point iterations x 25 SOR iterations with DeepFlow's level parameters"""
    i0, i1 = _textured_pair(w, h, 7 * w + h)
    rng = np.random.default_rng(w + 1000 * h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 5, 25, a, d, g, 1.6)
    try:
        fs_big.set_sor_variant(5, 5, 64, 64)
        gu, gv = fs_big.varref_f32(np.stack([i0, i0]), np.stack([i1, i1]), np.stack([u0, u0]), np.stack([v0, v0]), 5, 25, a, d, g, 1.6)
    finally:
        fs_big.set_sor_variant()
    for b in range(2):
        assert np.array_equal(gu[b].view(np.uint32), ou.view(np.uint32)), (w, h, b, float(np.abs(gu[b] - ou).max()))
        assert np.array_equal(gv[b].view(np.uint32), ov.view(np.uint32)), (w, h, b, float(np.abs(gv[b] - ov).max()))


# k_sor_wave: one, two, three and more column strips (128 columns per wave; kept widths of 118 / 108 next to a cut), odd widths (a last lane with one pixel inside the image),
# a level lower than the 10-row halo, and 1 / 2 / 3 / 5 row bands (a band cut costs 10 rows of halo on either side; bands lower than the halo)
WAVE_SHAPES = [(129, 67, 1), (128, 70, 2), (127, 66, 3), (236, 40, 1), (237, 36, 2), (230, 173, 2), (230, 173, 5), (302, 73, 3), (303, 65, 1), (385, 288, 3), (768, 431, 2), (1000, 9, 1), (1001, 9, 2), (612, 99, 5)]


@pytest.mark.parametrize("w,h,bands", WAVE_SHAPES)
def test_wave_solver_equals_the_oracle_on_strip_and_band_cuts(fs_big, w, h, bands):
    """the one-wave row pipeline (k_sor_wave, solver mode 6 = on every level beyond one workgroup) against the ORACLE's VariationalRefinement: 5 fixed-point iterations x 25 SOR
    iterations with DeepFlow's level parameters, every column-strip and row-band cut"""
    i0, i1 = _textured_pair(w, h, 7 * w + h)
    rng = np.random.default_rng(w + 1000 * h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 5, 25, a, d, g, 1.6)
    try:
        fs_big.set_sor_variant(6, 5, 64, 64); fs_big.set_wave_solver(True, 0, bands)
        gu, gv = fs_big.varref_f32(np.stack([i0, i0]), np.stack([i1, i1]), np.stack([u0, u0]), np.stack([v0, v0]), 5, 25, a, d, g, 1.6)
    finally:
        fs_big.set_sor_variant(); fs_big.set_wave_solver()
    for b in range(2):
        assert np.array_equal(gu[b].view(np.uint32), ou.view(np.uint32)), (w, h, b, float(np.abs(gu[b] - ou).max()), int((gu[b] != ou).sum()))
        assert np.array_equal(gv[b].view(np.uint32), ov.view(np.uint32)), (w, h, b, float(np.abs(gv[b] - ov).max()))


def test_wave_solver_on_random_level_sizes_equals_the_per_colour_kernel(fs_big):
    """40 random level sizes (129 - 1500 columns: one to fourteen strips; 9 - 432 rows) and band counts (automatic, 1 - 6): the one-wave pipelines (mode 6) against one launch
    per colour and iteration (mode 0, the kernel the oracle tests pin on the fixed shapes above) -- every float of both pairs bit-identical"""
    rng = np.random.default_rng(20260505)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    done = 0
    while done < 40:
        w = int(rng.integers(129, 1501)); h = int(rng.integers(9, 433))
        if w * h <= 8192 or w * h > 768 * 432:
            continue
        bands = int(rng.integers(0, 7))
        i0, i1 = _textured_pair(w, h, 3 * w + h)
        u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
        args = (np.stack([i0, i1]), np.stack([i1, i0]), np.stack([u0, v0]), np.stack([v0, u0]), 2, 10, a, d, g, 1.6)
        try:
            fs_big.set_sor_variant(0, 5, 64, 64); ru, rv = fs_big.varref_f32(*args)
            fs_big.set_sor_variant(6, 5, 64, 64); fs_big.set_wave_solver(True, 0, bands); gu, gv = fs_big.varref_f32(*args)
        finally:
            fs_big.set_sor_variant(); fs_big.set_wave_solver()
        assert np.array_equal(gu.view(np.uint32), ru.view(np.uint32)) and np.array_equal(gv.view(np.uint32), rv.view(np.uint32)), (w, h, bands, int((gu != ru).sum()))
        done += 1


def test_wave_solver_on_the_768x432_pyramid(fs_big):
    """DeepFlow on the 768 x 432 grid (57 levels): one-wave pipelines with automatic bands (mode 6) == one launch per colour (mode 0) on every pixel of both pairs"""
    a0, a1 = _textured_pair(768, 432, 5); b0, b1 = _textured_pair(768, 432, 6)
    i0 = np.stack([a0, b1]).astype(np.uint8); i1 = np.stack([a1, b0]).astype(np.uint8)
    try:
        fs_big.set_sor_variant(0, 5, 64, 64); ru, rv = fs_big.deepflow(i0, i1)
        fs_big.set_sor_variant(6, 5, 64, 64); u, v = fs_big.deepflow(i0, i1)
    finally:
        fs_big.set_sor_variant()
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32))


def test_streaming_solver_on_the_768x432_pyramid(fs_big):
    """DeepFlow on the 1280 x 720 configuration's flow grid (768 x 432, 57 levels, six column strips of 128 on the top level): streaming (mode 5) ==
    one launch per colour (mode 0) on every pixel of both pairs"""
    assert len(fs_big.levels()) == 57 and fs_big.levels()[0] == (768, 432)
    a0, a1 = _textured_pair(768, 432, 5); b0, b1 = _textured_pair(768, 432, 6)
    i0 = np.stack([a0, b1]).astype(np.uint8); i1 = np.stack([a1, b0]).astype(np.uint8)
    try:
        fs_big.set_sor_variant(0, 5, 64, 64); ru, rv = fs_big.deepflow(i0, i1)
        fs_big.set_sor_variant(5, 5, 64, 64); u, v = fs_big.deepflow(i0, i1)
    finally:
        fs_big.set_sor_variant()
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(v.view(np.uint32), rv.view(np.uint32))
    assert np.abs(ru).max() > 0.5                   # a real flow field, not zeros


# widths around the 60-column tiles of k_coef_lanes (one tile with idle lanes, a last tile of 1, 2, 3 columns, a right border inside the two feeding lanes) and heights around its
# 4-row groups / 16-row workgroups (a last group of 1, 2, 3 rows, images lower than the 5-row vertical stencil)
COEF_SHAPES = [(7, 3), (33, 26), (58, 5), (59, 17), (60, 16), (61, 15), (62, 18), (63, 33), (64, 4), (120, 35), (121, 9), (122, 64), (123, 65), (181, 30), (240, 66), (384, 288), (768, 432)]


@pytest.mark.parametrize("w,h", COEF_SHAPES)
def test_coefficients_from_lanes_equal_coefficients_from_memory(fs_big, w, h):
    """k_coef_lanes (neighbours through whole-wave shifts, rows in registers) against k_coef (every neighbour loaded with clamped indices): the refinement's flow is
    bit-identical; the small sizes also against the oracle"""
    import ctypes as C
    from sindslam_amd._lib import lib
    i0, i1 = _textured_pair(w, h, 3 * w + h)
    rng = np.random.default_rng(w + 977 * h)
    u0 = rng.normal(0, 1.5, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.5, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    args = (np.stack([i0, i1]), np.stack([i1, i0]), np.stack([u0, v0]), np.stack([v0, u0]), 3, 7, a, d, g, 1.6)
    try:
        fs_big.set_coef_kernel(0); mu, mv = fs_big.varref_f32(*args)
        fs_big.set_coef_kernel(2); iu, iv = fs_big.varref_f32(*args)          # lanes, the compiler's IEEE sqrt and division
        fs_big.set_coef_kernel(1); lu, lv = fs_big.varref_f32(*args)          # lanes, short forms (the default)
    finally:
        fs_big.set_coef_kernel(1)
    assert lib().sind_flow_set_coef_kernel(fs_big._h, 4) == -1
    assert np.array_equal(iu.view(np.uint32), mu.view(np.uint32)) and np.array_equal(iv.view(np.uint32), mv.view(np.uint32)), (w, h, float(np.abs(iu - mu).max()))
    assert np.array_equal(lu.view(np.uint32), mu.view(np.uint32)) and np.array_equal(lv.view(np.uint32), mv.view(np.uint32)), (w, h, float(np.abs(lu - mu).max()))
    assert np.isfinite(lu).all() and np.abs(lu - np.stack([u0, v0])).max() > 1e-3          # the refinement moved the field
    if w * h <= 130 * 70:
        ou, ov = O.varref(i0, i1, u0, v0, 3, 7, a, d, g, 1.6)
        assert np.array_equal(lu[0].view(np.uint32), ou.view(np.uint32)) and np.array_equal(lv[0].view(np.uint32), ov.view(np.uint32))


@pytest.mark.parametrize("w,h,cap", [(384, 287, 2), (300, 101, 1), (153, 99, 3), (129, 67, 1)])
def test_persistent_solver_workgroups_equal_one_workgroup_per_item(fs_big, w, h, cap):
    """sind_flow_set_solver_workgroups: `cap` workgroups walk the (strip, image) items of a streaming launch in turn (2 images x 1-3 strips here: 2-6 items, so a workgroup takes
    two, three or all six of them, and the LDS of an item is cleared behind a barrier) -- the flow equals the oracle's bit for bit"""
    from sindslam_amd._lib import lib
    i0, i1 = _textured_pair(w, h, 5 * w + h)
    rng = np.random.default_rng(w + 31 * h)
    u0 = rng.normal(0, 1.0, (h, w)).astype(np.float32); v0 = rng.normal(0, 1.0, (h, w)).astype(np.float32)
    a, d, g = 4 * np.float32(1.0), np.float32(0.5) / np.float32(3), np.float32(5.0) / np.float32(3)
    ou, ov = O.varref(i0, i1, u0, v0, 2, 10, a, d, g, 1.6); ou2, ov2 = O.varref(i1, i0, v0, u0, 2, 10, a, d, g, 1.6)
    try:
        fs_big.set_sor_variant(5, 5, 64, 64); fs_big.set_solver_workgroups(cap)
        gu, gv = fs_big.varref_f32(np.stack([i0, i1]), np.stack([i1, i0]), np.stack([u0, v0]), np.stack([v0, u0]), 2, 10, a, d, g, 1.6)
    finally:
        fs_big.set_sor_variant(); fs_big.set_solver_workgroups(0)
    assert lib().sind_flow_set_solver_workgroups(fs_big._h, -1) == -1
    assert np.array_equal(gu[0].view(np.uint32), ou.view(np.uint32)) and np.array_equal(gv[0].view(np.uint32), ov.view(np.uint32))
    assert np.array_equal(gu[1].view(np.uint32), ou2.view(np.uint32)) and np.array_equal(gv[1].view(np.uint32), ov2.view(np.uint32))


def test_short_forms_of_sqrt_and_quotient_are_exact():
    """k_coef_lanes' sqrt (hardware estimate + the neighbour its residual asks for, without the guards for tiny / zero / infinite arguments) and c / b through the reciprocal
    against sqrtf and the IEEE division: every one of the 2^23 significands x the binary exponents -40..40 (the kernel's arguments are >= epsilon^2 = 2^-20, its roots lie in
    [2^-10, 2^14]), numerators = alpha / 2, delta / 2, gamma / 2 of DeepFlow's levels and of the refinement's defaults -- not one differs"""
    import ctypes as C
    from sindslam_amd._lib import check, lib
    for numer in ((2.0, np.float32(0.5) / np.float32(3) / 2, np.float32(5.0) / np.float32(3) / 2), (10.0, 2.5, 5.0)):
        out = (C.c_ulonglong * 2)(); n3 = (C.c_float * 3)(*[float(x) for x in numer])
        check(lib().sind_debug_coef_math_scan(0, -40, 40, n3, out), "sind_debug_coef_math_scan")
        assert out[0] == 0 and out[1] == 0, (numer, out[0], out[1])
    out = (C.c_ulonglong * 2)(); n3 = (C.c_float * 3)(1.0, 3.0, 7.0)
    check(lib().sind_debug_coef_math_scan(0, -96, -41, n3, out), "sind_debug_coef_math_scan"); assert out[0] == 0          # the square root down to the first argument it is made for
    assert lib().sind_debug_coef_math_scan(0, -97, 0, n3, out) == -1


def test_division_through_the_reciprocal_is_exact():
    """the solver's division (hardware reciprocal + one Newton step, then Markstein's correction) against the IEEE division: every one of the 2^23
    float significands, binary exponents -24..24 (the step is scale invariant; the system's diagonal lies in [0.01, 1e4]), 16 numerators per divisor,
    quotients from 2^-40 to 2^40 (k_coef divides squared image derivatives by gradient norms >= 0.01 the same way) -- no reciprocal and no quotient differs"""
    import ctypes as C
    from sindslam_amd._lib import check, lib
    out = (C.c_ulonglong * 3)()
    check(lib().sind_debug_rcp_scan(0, -24, 24, out), "sind_debug_rcp_scan")
    assert out[0] == 0 and out[1] == 0, (out[0], out[1], hex(out[2]))
