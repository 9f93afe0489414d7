"""CPU: the product's HOST stages (libsind_host.so = sindslam_amd/csrc/host/*, built with plain g++) against the oracle.
These stages are serial by design (DESIGN.md "host stages") and are written independently of the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    so = os.path.join(ROOT, "sindslam_amd", "libsind_host.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sindslam_amd", "csrc"), "../libsind_host.so"])
    return C.CDLL(so)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def blobs(seed, density=0.02):
    rng = np.random.default_rng(seed)
    m = (rng.random((480, 640)) < density).astype(np.uint8) * 255
    for _ in range(12):
        x, y, w, h = rng.integers(0, 600), rng.integers(0, 440), rng.integers(5, 120), rng.integers(5, 90)
        m[y:y + h, x:x + w] = 255
    m[0:5, 0:7] = 255; m[470:480, 630:640] = 255
    return m


@pytest.mark.parametrize("n", [3, 4, 5, 7, 9, 10, 15])
def test_bit_morphology_matches_oracle(H, n):
    m = blobs(n)
    for op, code in (("dilate", 0), ("erode", 1), ("open", 2), ("close", 3)):
        a = np.zeros_like(m); H.sindh_morph(P(m), 640, 480, n, code, P(a))
        assert np.array_equal(a, O.morph(m, n, op)), (n, op)


@pytest.mark.parametrize("external", [1, 0])
def test_contours_match_oracle(H, external):
    m = O.morph(blobs(77, 0.002), 5, "close")
    m[100:160, 100:200] = 255; m[120:140, 130:170] = 0; m[125:135, 140:160] = 255      # nested hole / island
    pts = np.zeros((400000, 2), np.int32); lens = np.zeros(20000, np.int32)
    n = H.sindh_find_contours(P(m), 640, 480, external, P(pts), len(pts), P(lens), len(lens))
    ref = O.find_contours(m, bool(external))
    assert n == len(ref) and lens[:n].tolist() == [len(c) for c in ref]
    assert np.array_equal(pts[:lens[:n].sum()], np.concatenate(ref))


def test_drawing_matches_oracle_pipeline(H):
    m = O.morph(blobs(5, 0.0), 3, "open")
    for filled in (1, 0):
        a = np.zeros_like(m); H.sindh_draw(P(m), 640, 480, filled, P(a))
        if filled:      # filled external contours = the blobs with their holes closed; every source pixel is covered
            assert np.all(a[m > 0] == 255)
        else:           # thickness 2 = border pixels dilated by the 3x3 cross
            cs = O.find_contours(m, True); b = np.zeros_like(m)
            for c in cs:
                for dx, dy in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
                    x = np.clip(c[:, 0] + dx, 0, 639); y = np.clip(c[:, 1] + dy, 0, 479); b[y, x] = 255
            assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,n,outlier_step,noise", [(11, 2501, 5, 0.3), (12, 2902, 3, 1.0), (13, 403, 2, 0.1), (14, 7, 0, 0.0), (15, 2900, 0, 2.5)])
def test_homography_identical_to_oracle_more_cases(H, seed, n, outlier_step, noise):
    """pair counts that are not a multiple of the SIMD width, heavy outliers, few points, pure noise"""
    rng = np.random.default_rng(seed); Hm = np.array([[1.01, -0.02, 3.0], [0.015, 0.99, -2.5], [1e-6, -2e-6, 1.0]])
    src = rng.uniform(0, 640, (n, 2)).astype(np.float32)
    p = np.c_[src, np.ones(len(src))] @ Hm.T; dst = (p[:, :2] / p[:, 2:] + rng.normal(0, noise, (len(src), 2))).astype(np.float32)
    if outlier_step:
        dst[::outlier_step] += rng.uniform(-40, 40, dst[::outlier_step].shape).astype(np.float32)
    out = np.zeros(9)
    ok = H.sindh_find_homography(P(src), P(dst), len(src), P(out))
    rok, ref = O.find_homography(src, dst)
    assert bool(ok) == bool(rok) and np.array_equal(out.reshape(3, 3), ref)


def test_homography_identical_to_oracle(H):
    rng = np.random.default_rng(3); Hm = np.array([[0.99, 0.01, 4.0], [-0.01, 1.0, 1.5], [2e-6, 1e-6, 1.0]])
    src = rng.uniform(0, 640, (2500, 2)).astype(np.float32)
    p = np.c_[src, np.ones(len(src))] @ Hm.T; dst = (p[:, :2] / p[:, 2:] + rng.normal(0, 0.3, (len(src), 2))).astype(np.float32)
    dst[::7] += 25
    out = np.zeros(9)
    assert H.sindh_find_homography(P(src), P(dst), len(src), P(out)) == 1
    ok, ref = O.find_homography(src, dst)
    assert ok and np.array_equal(out.reshape(3, 3), ref)                         # same specification, same bits


def test_dilation_hit_test_equals_the_dilation(H):
    """BitImg::dilation_hits (the element's window around a query point, read from the source) == dilated().get at that point: sparse and dense random images, the 10 x 10
    ellipse of the CalOccluded contour filter and others, query points everywhere including the borders and corners"""
    rng = np.random.default_rng(5)
    for w, h, dens, elem in ((640, 480, 0.0005, 10), (640, 480, 0.01, 10), (128, 96, 0.02, 7), (192, 64, 0.3, 4), (64, 40, 0.002, 15), (70, 33, 0.05, 3)):
        img = (rng.random((h, w)) < dens).astype(np.uint8) * 255
        pts = np.stack([rng.integers(0, w, 4000), rng.integers(0, h, 4000)], axis=1).astype(np.int32)
        pts[:8] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1], [w // 2, 0], [0, h // 2], [w - 1, h // 2], [w // 2, h - 1]]
        pts = np.ascontiguousarray(pts)
        assert H.sindh_dilation_hits_check(P(np.ascontiguousarray(img)), w, h, elem, P(pts), len(pts)) == 0, (w, h, dens, elem)


def test_four_lane_plane_fit_equals_scalar(H):
    """peac_fit4 (the merge candidates' fits, four in the lanes of an AVX2 register) == peac_fit (one at a time, the function the device and the oracle's restatement use):
    centre, normal and MSE bit for bit -- random point sets, exactly planar and axis-aligned sets (zero pivots: rotations skipped per lane), lanes that converge after
    different numbers of sweeps, groups of 1-3"""
    rng = np.random.default_rng(11)
    sets = []
    for k in range(403):
        n = int(rng.integers(4, 400)); kind = k % 6
        if kind == 0: pts = rng.normal(0, 1, (n, 3)) * rng.uniform(0.01, 3, 3) + rng.normal(0, 2, 3)
        elif kind == 1: pts = np.c_[rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.full(n, 1.5)]                      # exactly planar, axis aligned
        elif kind == 2: xy = rng.uniform(-1, 1, (n, 2)); pts = np.c_[xy, 0.3 * xy[:, 0] - 0.2 * xy[:, 1] + 2 + rng.normal(0, 1e-3, n)]   # a tilted wall with depth noise
        elif kind == 3: pts = np.repeat(rng.normal(0, 1, (1, 3)), n, 0)                                                  # one point n times: the zero matrix
        elif kind == 4: pts = np.c_[rng.uniform(-1, 1, n), np.zeros(n), np.zeros(n)]                                     # a line along x
        else: pts = rng.normal(0, 1, (n, 3)) * np.array([1.0, 1.0, 1e-6])
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        sets.append(([x.sum(), y.sum(), z.sum(), (x * x).sum(), (y * y).sum(), (z * z).sum(), (x * y).sum(), (y * z).sum(), (x * z).sum()], n))
    m = np.ascontiguousarray([s[0] for s in sets], np.float64); cnt = np.ascontiguousarray([s[1] for s in sets], np.int32)
    a = np.zeros((len(sets), 7)); b = np.zeros((len(sets), 7))
    H.sindh_peac_fits(P(m), P(cnt), len(sets), 0, P(a)); H.sindh_peac_fits(P(m), P(cnt), len(sets), 1, P(b))
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), np.argwhere(a.view(np.uint64) != b.view(np.uint64))[:5]
    assert np.isfinite(a[::6]).all() and np.abs(np.linalg.norm(a[::6, 3:6], axis=1) - 1).max() < 1e-12


def test_peac_plane_contours_identical(H, frames):
    from sindslam_amd.synth import TUM3
    bgr, depth = frames
    dd = O.DynaDetect(bgr[1], bgr[0], TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    dd.detect(bgr[2], depth[2])
    out = np.zeros((480, 640), np.uint8); d = np.ascontiguousarray(depth[2])
    H.sindh_peac(P(d), 640, 480, C.c_float(TUM3["fx"]), C.c_float(TUM3["fy"]), C.c_float(TUM3["cx"]), C.c_float(TUM3["cy"]), C.c_float(5000.0), P(out))
    assert np.array_equal(out, dd.debug()["plane_contours"]) and out.any()


def test_octree_identical(H, frames):
    bgr, _ = frames; orb = O.ORBextractor(1500, 1.2, 8, 15, 5); orb.extract(O.bgr2gray(bgr[2]))
    per = orb.tables()["per_level"]
    for lv in range(8):
        fk = orb.fast_keypoints(lv); sel = orb.selected(lv); w, h = orb.level_size(lv)
        xyr = np.stack([fk["x"], fk["y"], fk["response"]], 1).astype(np.float32).copy(); out = np.zeros((4000, 3), np.float32)
        n = H.sindh_octree(P(xyr), len(xyr), 16, w - 16, 16, h - 16, int(per[lv]), P(out), 4000)
        assert n == len(sel)
        assert np.array_equal(out[:n, 0] + 16, sel["x"]) and np.array_equal(out[:n, 1] + 16, sel["y"]) and np.array_equal(out[:n, 2], sel["response"])


def _blobs(rng, w, h, n):
    img = np.zeros((h, w), np.uint8)
    for _ in range(n):
        cx, cy, rx, ry = rng.integers(0, w), rng.integers(0, h), rng.integers(3, 60), rng.integers(3, 30)
        y0, y1, x0, x1 = max(cy - ry, 0), min(cy + ry, h - 1), max(cx - rx, 0), min(cx + rx, w - 1)
        img[y0:y1 + 1, x0:x1 + 1] = (rng.random((y1 - y0 + 1, x1 - x0 + 1)) < 0.9) * 255
    return img


@pytest.mark.parametrize("w", [640, 200, 64])
def test_window_restricted_morphology_equals_full_frame_oracle(H, w):
    """eroded_rows / opened_rows / dilated(rows) evaluate only the mask's own rows and words; the result must equal the full-frame op"""
    import oracle_lib as O
    rng = np.random.default_rng(w)
    for trial in range(6):
        img = _blobs(rng, w, 90, 1 + trial % 3)
        for n in (3, 4, 7, 9, 10):
            for op_rows, op_full in ((4, "erode"), (5, "open"), (6, "dilate")):
                out = np.zeros_like(img); H.sindh_morph_rows(P(img), w, 90, n, op_rows, P(out))
                assert np.array_equal(out, O.morph(img, n, op_full)), (trial, n, op_rows)


def test_span_flood_fill_equals_connected_component(H):
    from scipy import ndimage
    rng = np.random.default_rng(5)
    for trial in range(40):
        w, h = int(rng.choice([64, 100, 640])), 70
        same = ((rng.random((h, w)) < rng.uniform(0.4, 0.8)) * 255).astype(np.uint8); blocked = ((rng.random((h, w)) < 0.15) * 255).astype(np.uint8)
        sx, sy = int(rng.integers(0, w)), int(rng.integers(0, h))
        filled = np.zeros_like(same); bl2 = np.zeros_like(same)
        area = H.sindh_flood_fill(P(same), P(blocked), w, h, sx, sy, P(filled), P(bl2))
        if blocked[sy, sx]:
            assert area == 0 and not filled.any(); continue
        allowed = (same > 0) & (blocked == 0); allowed[sy, sx] = True          # the seed is filled unconditionally
        lab, _ = ndimage.label(allowed, structure=np.ones((3, 3)))
        want = lab == lab[sy, sx]
        assert np.array_equal(filled > 0, want) and area == want.sum() and np.array_equal(bl2 > 0, (blocked > 0) | want)


def test_sse_bit_packing_round_trip(H):
    rng = np.random.default_rng(9)
    for w in (640, 100, 64, 200):
        img = rng.choice(np.array([0, 7, 125, 255], np.uint8), size=(37, w))
        a = np.zeros_like(img); b = np.zeros_like(img)
        H.sindh_pack_roundtrip(P(img), w, 37, 125, P(a), P(b))
        assert np.array_equal(a, np.where(img > 0, 200, 3)) and np.array_equal(b, np.where(img == 125, 255, 0))
