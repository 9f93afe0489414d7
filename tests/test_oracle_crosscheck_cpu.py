"""CPU: the oracle's OpenCV restatements against INDEPENDENT implementations of the same primitive (scipy in this interpreter,
scikit-image 0.18 in the image's conda interpreter through tests/crosscheck_skimage.py).  The reference ships no golden vectors and
OpenCV is not installed ("parity unpinned"), so these are secondary evidence only: where scipy / scikit-image define
the operation exactly like OpenCV, the oracle must agree with them."""
import os
import subprocess

import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O

CONDA_PY = "/opt/conda/bin/python3.9"
HERE = os.path.dirname(os.path.abspath(__file__))


def _blobs(seed, h=120, w=160):
    rng = np.random.default_rng(seed)
    m = (rng.random((h, w)) < 0.03).astype(np.uint8) * 255
    for _ in range(8):
        x, y, bw, bh = rng.integers(0, w - 10), rng.integers(0, h - 10), rng.integers(3, 50), rng.integers(3, 40)
        m[y:y + bh, x:x + bw] = 255
    return m


@pytest.mark.parametrize("n", [3, 4, 5, 7, 9, 10, 15])
def test_morphology_equals_scipy_rank_filters(n):
    """cv::dilate / erode with an n x n element anchored at (n/2, n/2), border = "ignore" <=> scipy maximum/minimum_filter with that
    footprint (correlation order, centre n // 2) and constant border 0 / 255.  The footprint is read back from the oracle itself
    (response to a single pixel, point-mirrored), so this checks anchor, orientation and border handling, not the ellipse formula."""
    single = np.zeros((41, 41), np.uint8); single[20, 20] = 255
    d = O.morph(single, n, "dilate") > 0
    a = n // 2
    se = np.zeros((n, n), bool)
    for i in range(n):
        for j in range(n):
            se[i, j] = d[20 - (i - a), 20 - (j - a)]
    assert se.any() and d.sum() == se.sum()
    for seed in range(3):
        m = _blobs(seed)
        assert np.array_equal(O.morph(m, n, "dilate"), ndimage.maximum_filter(m, footprint=se, mode="constant", cval=0))
        assert np.array_equal(O.morph(m, n, "erode"), ndimage.minimum_filter(m, footprint=se, mode="constant", cval=255))
        op = ndimage.maximum_filter(ndimage.minimum_filter(m, footprint=se, mode="constant", cval=255), footprint=se, mode="constant", cval=0)
        assert np.array_equal(O.morph(m, n, "open"), op)


def test_median5_equals_scipy():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 40000, (97, 131)).astype(np.float32); img[rng.random(img.shape) < 0.1] = 0
    assert np.array_equal(O.median5_f32(img), ndimage.median_filter(img, size=5, mode="nearest"))          # BORDER_REPLICATE


def test_gaussian_blurs_match_float_convolution():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (90, 120)).astype(np.uint8)
    # 7x7, sigma 2 on u8 (ORB), BORDER_REFLECT_101.  OpenCV 4.2 filters u8 in 8.8 fixed point with every tap rounded on its own: the taps
    # are (18 34 49 55 49 34 18) / 256, which sum to 257/256, so the result sits up to ~2 grey levels above the exact float convolution
    # and within one level of the float convolution with the rounded taps
    x = np.arange(-3, 4, dtype=np.float64); k = np.exp(-x * x / (2 * 2.0 * 2.0)); k /= k.sum()
    got = O.gaussian_blur_u8(img, 7, 2.0).astype(np.float64)
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    assert np.abs(got - ref).max() <= 2.5
    kq = np.round(k * 256) / 256; assert kq.sum() * 256 == 257
    refq = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), kq, axis=0, mode="mirror"), kq, axis=1, mode="mirror")
    assert np.abs(got - np.minimum(refq, 255)).max() <= 1.0
    # 3 taps, sigma 0.6 on f32 (DeepFlow's pyramid base)
    x = np.arange(-1, 2, dtype=np.float64); k = np.exp(-x * x / (2 * 0.6 * 0.6)); k /= k.sum()
    f = img.astype(np.float32)
    ref = ndimage.correlate1d(ndimage.correlate1d(f.astype(np.float64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    assert np.abs(O.gaussian_blur3_f32(f, 0.6) - ref).max() < 1e-3


def test_gray_and_bilinear_resize_match_float_formulas():
    rng = np.random.default_rng(6)
    bgr = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
    ref = 0.114 * bgr[..., 0] + 0.587 * bgr[..., 1] + 0.299 * bgr[..., 2]
    assert np.abs(O.bgr2gray(bgr).astype(np.float64) - ref).max() <= 1.0
    # INTER_LINEAR: source coordinate (x + 0.5) * scale - 0.5, clamped to the image
    src = rng.uniform(0, 255, (60, 80)).astype(np.float32)
    for dh, dw in ((57, 76), (36, 48), (100, 133)):
        yy, xx = np.meshgrid((np.arange(dh) + 0.5) * (60 / dh) - 0.5, (np.arange(dw) + 0.5) * (80 / dw) - 0.5, indexing="ij")
        ref = ndimage.map_coordinates(src.astype(np.float64), [yy, xx], order=1, mode="nearest")
        assert np.abs(O.resize_f32(src, dw, dh) - ref).max() < 2e-3, (dh, dw)
        u8 = src.astype(np.uint8)
        ref8 = ndimage.map_coordinates(u8.astype(np.float64), [yy, xx], order=1, mode="nearest")
        assert np.abs(O.resize_u8(u8, dw, dh).astype(np.float64) - ref8).max() <= 1.0, (dh, dw)


def test_external_contours_are_the_8_connected_components():
    for seed in range(4):
        m = _blobs(20 + seed)
        lab, n = ndimage.label(m > 0, structure=np.ones((3, 3)))
        cs = O.find_contours(m, True)
        assert len(cs) == n
        seen = set()
        for c in cs:
            ids = set(lab[c[:, 1], c[:, 0]].tolist()); assert len(ids) == 1 and 0 not in ids      # a contour stays on one component
            cid = ids.pop(); assert cid not in seen; seen.add(cid)
            ys, xs = np.nonzero(lab == cid); first = np.lexsort((xs, ys))[0]
            assert c[0].tolist() == [xs[first], ys[first]]                                        # starts at the raster-first pixel
        assert seen == set(range(1, n + 1))
        # without holes, the traced pixels are exactly the inner boundary: foreground pixels with a background (or outside) 4-neighbour
        filled = ndimage.binary_fill_holes(m > 0, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]]))
        inner = filled & ~ndimage.binary_erosion(filled, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]]), border_value=0)
        traced = np.zeros_like(filled)
        for c in O.find_contours(filled.astype(np.uint8) * 255, True):
            traced[c[:, 1], c[:, 0]] = True
        assert np.array_equal(traced, inner)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="the image's conda interpreter (scikit-image) is not present")
def test_fast_scores_otsu_and_resize_against_scikit_image(frames, tmp_path):
    bgr, _ = frames
    gray = O.bgr2gray(bgr[2])
    orb = O.ORBextractor(1500, 1.2, 8, 15, 5); orb.extract(gray)
    lv = 1; pad = orb.level_padded(lv); fk = orb.fast_keypoints(lv)
    rng = np.random.default_rng(11)
    otsu_imgs = []
    for k in range(6):            # bimodal / skewed images
        a = rng.normal(60 + 10 * k, 12 + k, (80, 100)); b = rng.normal(170 - 5 * k, 20, (80, 100)); sel = rng.random((80, 100)) < (0.2 + 0.1 * k)
        otsu_imgs.append(np.clip(np.where(sel, b, a), 0, 255).astype(np.uint8))
    otsu_imgs = np.stack(otsu_imgs)
    src = rng.uniform(0, 255, (60, 80)).astype(np.float32)
    kps, desc = orb.extract(gray)
    l0 = kps["octave"] == 0; k0 = kps[l0]; d0 = desc[l0]
    blurred = O.gaussian_blur_u8(gray, 7, 2.0)             # ORBextractor.cc:1145: the level image is cloned, then blurred
    fin, fout = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(fin, fast_img=pad, otsu_imgs=otsu_imgs, resize_src=src, resize_shape=np.array([57, 76]), brief_img=blurred,
             brief_rc=np.stack([np.rint(k0["y"]), np.rint(k0["x"])], 1).astype(np.int64), brief_angle=np.deg2rad(k0["angle"].astype(np.float64)),
             angle_img=orb.level_padded(0), angle_rc=np.stack([np.rint(k0["y"]) + 19, np.rint(k0["x"]) + 19], 1).astype(np.int64))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    subprocess.check_call([CONDA_PY, os.path.join(HERE, "crosscheck_skimage.py"), fin, fout], env=env, timeout=600)
    r = np.load(fout)
    # FAST: every cell-wise keypoint of the oracle (position relative to the 16-px border, image padded by 19) is a FAST-9 corner whose
    # OpenCV score (largest threshold at which it is still a corner) equals scikit-image's segment test evaluated threshold by threshold
    assert len(fk) > 200
    px = fk["x"].astype(int) + 35; py = fk["y"].astype(int) + 35
    s = r["score"][py, px]
    assert np.array_equal(s, fk["response"].astype(np.int16)), float(np.mean(s == fk["response"]))
    assert s.min() >= 5
    # non-maximum suppression: a kept corner beats its 8 neighbours strictly -- except where the neighbour lies outside the 30-px cell
    # view the corner was found in (ORB runs FAST cell by cell), which concerns a few per cent of the keypoints
    nb = np.stack([r["score"][py + dy, px + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)])
    assert np.mean((nb < s[None, :]).all(0)) > 0.93
    # Otsu: same class split as cv::threshold(THRESH_OTSU); Triangle: OpenCV's variant ends with an extra "thresh--" and its own flip rule
    for k, im in enumerate(otsu_imgs):
        hist = np.bincount(im.ravel(), minlength=256)
        assert O.otsu(hist) == r["otsu"][k], k
        assert abs(O.triangle(hist) - r["triangle"][k]) <= 2, (k, O.triangle(hist), r["triangle"][k])
    assert np.abs(O.resize_f32(src, 76, 57) - r["resized"]).max() < 2e-3
    # rBRIEF: the 256 x 4 pattern table is scikit-image's copy of OpenCV's; steering it by the oracle's keypoint angle over the oracle's
    # blurred image reproduces the oracle's level-0 descriptors (a handful of bits may differ where a rotated offset falls on x.5:
    # C round() vs cvRound, and float vs double trigonometry)
    import re
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "sind_brief_pattern.h")).read()
    table = np.array([int(v) for v in re.findall(r"-?\d+", hdr[hdr.index("{"):])], np.int32).reshape(256, 4)
    assert np.array_equal(table, r["pattern"])
    assert len(k0) > 300
    diff_bits = np.unpackbits(d0 ^ r["brief"], axis=1).sum()
    assert diff_bits <= 2e-4 * d0.size * 8, (int(diff_bits), d0.size * 8)
    # IC_Angle: atan2 of the first-order moments over the radius-15 disc; the oracle goes through cv::fastAtan2 (about 0.3 degrees accurate)
    da = np.abs((np.rad2deg(r["angle"]) % 360.0) - k0["angle"]); da = np.minimum(da, 360.0 - da)
    assert da.max() < 0.5, float(da.max())
