"""CPU: properties of the oracle's restatements of the OpenCV pieces that NO fixture of the reference pins (parity unpinned) and that
carry the path -- the variational flow (99 % of the bytes), cv::kmeans with KMEANS_USE_INITIAL_LABELS, floodFill(MASK_ONLY).  These do not
depend on the details of the restatement: they are what any correct implementation of the published algorithms must satisfy."""
import ctypes as C

import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O


def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    t = ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0) * 3 + ndimage.gaussian_filter(rng.standard_normal((h, w)), 6.0) * 8
    t = (t - t.min()) / (t.max() - t.min())
    return t * 200 + 25


@pytest.mark.timeout(300)
def test_deepflow_endpoint_error_on_affine_warps():
    """I1(A x) = I0(x): the recovered flow must be A x - x to well below a tenth of a pixel on the 384 x 288 flow grid (translation + rotation +
    scale, up to ~4 px of motion), for any correct Brox-type solver"""
    h, w = 288, 384; yy, xx = np.mgrid[0:h, 0:w].astype(np.float64); cx, cy = w / 2, h / 2
    base = _texture(h + 40, w + 40, 7)
    for ang, sc, tx, ty in ((0.0, 1.0, 2.3, -1.4), (0.006, 1.0, 0.5, 0.7), (-0.004, 1.008, -1.2, 0.4)):
        A = sc * np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        fx = A[0, 0] * (xx - cx) + A[0, 1] * (yy - cy) + cx + tx - xx; fy = A[1, 0] * (xx - cx) + A[1, 1] * (yy - cy) + cy + ty - yy
        I0 = ndimage.map_coordinates(base, [yy + 20, xx + 20], order=3)
        # I1(y) = I0(A^-1 (y - t)): sample the base texture at the back-warped positions
        Ai = np.linalg.inv(A); bx = Ai[0, 0] * (xx - cx - tx) + Ai[0, 1] * (yy - cy - ty) + cx; by = Ai[1, 0] * (xx - cx - tx) + Ai[1, 1] * (yy - cy - ty) + cy
        I1 = ndimage.map_coordinates(base, [by + 20, bx + 20], order=3)
        f = O.deepflow(np.clip(np.rint(I0), 0, 255).astype(np.uint8), np.clip(np.rint(I1), 0, 255).astype(np.uint8))
        epe = np.hypot(f[..., 0] - fx, f[..., 1] - fy)[16:-16, 16:-16]
        assert epe.mean() < 0.1 and np.percentile(epe, 99) < 0.35, (ang, sc, tx, ty, epe.mean(), np.percentile(epe, 99))


@pytest.mark.timeout(300)
def test_deepflow_flip_symmetry_and_refinement_does_not_hurt():
    """mirroring both images mirrors the flow (u changes sign); one more VariationalRefinement of a good flow keeps it good (a fixed point of the scheme)"""
    h, w = 144, 192
    base = _texture(h + 20, w + 20, 3); yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    I0 = np.clip(np.rint(ndimage.map_coordinates(base, [yy + 10, xx + 10], order=3)), 0, 255).astype(np.uint8)
    I1 = np.clip(np.rint(ndimage.map_coordinates(base, [yy + 10 - 0.8, xx + 10 - 1.7], order=3)), 0, 255).astype(np.uint8)      # I1(x + (1.7, 0.8)) = I0(x)
    f = O.deepflow(I0, I1); g = O.deepflow(np.ascontiguousarray(I0[:, ::-1]), np.ascontiguousarray(I1[:, ::-1]))
    inner = (slice(12, -12), slice(12, -12))
    assert abs(f[inner][..., 0].mean() - 1.7) < 0.05 and abs(f[inner][..., 1].mean() - 0.8) < 0.05
    assert np.abs(g[:, ::-1, 0] + f[..., 0])[inner].mean() < 0.03 and np.abs(g[:, ::-1, 1] - f[..., 1])[inner].mean() < 0.03
    ru, rv = O.varref(I0.astype(np.float32), I1.astype(np.float32), f[..., 0], f[..., 1])
    assert np.hypot(ru - 1.7, rv - 0.8)[inner].mean() < 0.08


def _kmeans(data, labels, K, max_count, eps):
    data = np.ascontiguousarray(data, np.float32); lab = np.ascontiguousarray(labels, np.int32).copy(); ctr = np.zeros((K, data.shape[1]), np.float32)
    O.lib().orc_kmeans(O._p(data), len(data), data.shape[1], K, O._p(lab), int(max_count), C.c_double(eps), O._p(ctr)); return lab, ctr


def test_kmeans_converged_state_is_a_fixed_point():
    """run to convergence (eps 0, many iterations): every point sits with its nearest centre and every centre is the mean of its members"""
    rng = np.random.default_rng(1); K = 5
    means = rng.uniform(-5, 5, (K, 3)); data = np.concatenate([m + rng.normal(0, 0.4, (300, 3)) for m in means]).astype(np.float32)
    lab, ctr = _kmeans(data, rng.integers(0, K, len(data)), K, 100, 0.0)
    d = ((data[:, None, :] - ctr[None]) ** 2).sum(-1)
    assert np.array_equal(lab, d.argmin(1))
    for k in range(K):
        assert np.allclose(ctr[k], data[lab == k].mean(0), atol=1e-4)
    assert len(np.unique(lab)) == K


def test_kmeans_last_iteration_and_empty_cluster_rule():
    """(EPS+COUNT, 4, 0.07) as DynaDetect.cc:319 uses it: at most three assignment passes, the centres of the LAST pass are returned while the
    labels still belong to the centres before it; an empty cluster receives the point of the biggest cluster that is farthest from that cluster's centre"""
    rng = np.random.default_rng(2)
    data = np.concatenate([rng.normal(0, 1, (50, 3)), rng.normal(8, 1, (30, 3))]).astype(np.float32)
    init = np.zeros(80, np.int32); init[50:] = 1                       # K = 3, cluster 2 starts empty
    lab, ctr = _kmeans(data, init, 3, 2, 1e9)                          # maxCount clamps to 2, huge eps: one centre pass + repair, one assignment, stop
    big = data[:50]; far = int(np.argmax(((big - big.mean(0)) ** 2).sum(1)))          # the farthest member of the biggest cluster (cluster 0)
    # after the single assignment pass every label is the nearest of the repaired first-pass centres
    c0 = (big.sum(0) - big[far]) / 49; c1 = data[50:].mean(0); c2 = big[far]
    first = np.stack([c0, c1, c2]).astype(np.float32)
    assert np.array_equal(lab, ((data[:, None, :] - first[None]) ** 2).sum(-1).argmin(1))
    assert lab[far] == 2 and (lab == 2).sum() >= 1
    # and the returned centres are the means under those labels (the last pass recomputes centres, then stops without re-assigning)
    for k in range(3):
        assert np.allclose(ctr[k], data[lab == k].mean(0), atol=1e-4)


def test_flood_fill_mask_only_is_the_connected_component():
    """DynaDetect.cc:1605: floodFill(maskLow (0 / 128), border mask, seed, 255, ..., lo = up = 5, 8 | FLOODFILL_MASK_ONLY | (50 << 8)) -- floating
    range, but on a two-valued image with a gap of 128 > 5 it fills exactly the 8-connected component of the seed's value outside the blocked mask"""
    rng = np.random.default_rng(3); h, w = 60, 80
    img = (ndimage.gaussian_filter(rng.standard_normal((h, w)), 3) > 0).astype(np.uint8) * 128
    blocked = (ndimage.gaussian_filter(rng.standard_normal((h, w)), 4) > 0.02)
    mask = np.full((h + 2, w + 2), 255, np.uint8); mask[1:-1, 1:-1] = np.where(blocked, 255, 0)
    ys, xs = np.nonzero((img == 128) & ~blocked); sy, sx = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    m = mask.copy()
    area = O.lib().orc_flood_fill_mask_only(O._p(np.ascontiguousarray(img)), w, h, O._p(m), sx, sy, 50, 5)
    lab, _ = ndimage.label((img == 128) & ~blocked, structure=np.ones((3, 3)))
    want = lab == lab[sy, sx]
    assert area == want.sum() and np.array_equal(m[1:-1, 1:-1] == 50, want)
    assert np.array_equal(m[1:-1, 1:-1][~want], mask[1:-1, 1:-1][~want])                  # nothing else touched
    # a blocked or out-of-image seed fills nothing
    m2 = mask.copy(); by, bx = np.nonzero(blocked); assert O.lib().orc_flood_fill_mask_only(O._p(np.ascontiguousarray(img)), w, h, O._p(m2), int(bx[0]), int(by[0]), 50, 5) == 0
    assert np.array_equal(m2, mask)
