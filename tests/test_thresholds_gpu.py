"""GPU: the one-wave residual-threshold kernel (k_flow_thresholds: Otsu + Triangle of cv::threshold and the clamping of DynaDetect.cc:1309-1367) against
its serial one-thread statement, the oracle's Otsu / Triangle, and a numpy FP64 walk of the one rounding chain that carries a division."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd._lib import check, lib, ptr

pytestmark = pytest.mark.gpu
W, H = 640, 480
N = W * H


def _hists(rng, n):
    """n histograms that sum to W*H, of the shapes the residual image produces and of the degenerate ones (257th word: float bits of the maximal residual)"""
    out = np.zeros((n, 257), np.int32)
    for k in range(n):
        kind = k % 10
        if kind == 0:      # residual-like: heavy head, long thin tail up to bin 255
            v = np.minimum(rng.gamma(1.5, rng.uniform(3, 40), N), 255).astype(np.int64); v[0] = 255
        elif kind == 1:    # two populations
            v = np.where(rng.random(N) < rng.uniform(0.05, 0.6), rng.normal(rng.uniform(120, 230), 12, N), rng.normal(rng.uniform(5, 60), 6, N)).clip(0, 255).astype(np.int64)
        elif kind == 2:    # everything in one bin
            v = np.full(N, rng.integers(0, 256), np.int64)
        elif kind == 3:    # two spikes
            a, b = rng.integers(0, 256, 2); v = np.where(rng.random(N) < rng.uniform(0.001, 0.999), a, b).astype(np.int64)
        elif kind == 4:    # uniform
            v = rng.integers(0, 256, N)
        elif kind == 5:    # a single pixel away from bin 0 (q2 just above / below the float epsilon test)
            v = np.zeros(N, np.int64); v[:rng.integers(1, 3)] = rng.integers(1, 256)
        elif kind == 6:    # peak at the right end (Triangle's flipped case)
            v = (255 - np.minimum(rng.gamma(2.0, rng.uniform(2, 30), N), 255)).astype(np.int64)
        elif kind == 7:    # sparse: a handful of occupied bins
            bins = rng.choice(256, rng.integers(2, 7), replace=False); v = rng.choice(bins, N)
        elif kind == 8:    # ties: equal peaks
            a, b, c = rng.choice(256, 3, replace=False); v = np.repeat([a, b, c], N // 3 + 1)[:N]
        else:              # narrow band in the middle
            lo = rng.integers(0, 200); v = rng.integers(lo, lo + rng.integers(2, 56), N)
        out[k, :256] = np.bincount(v, minlength=256)
        out[k, 256] = np.float32(rng.uniform(0.3, 80.0)).view(np.int32)
    return out


def _run(hist, variant, want_mu1=False):
    n = hist.shape[0]; res = np.zeros((n, 261), np.int32); mu1 = np.zeros((n, 256), np.float64) if want_mu1 else None
    check(lib().sind_debug_flow_thresholds(ptr(hist), n, W, H, variant, 0, ptr(res), ptr(mu1) if want_mu1 else None), "sind_debug_flow_thresholds")
    return res, mu1


def _mu1_chain(h):
    """numpy float64 = IEEE binary64, one operation at a time: the chain of cv::threshold's Otsu loop"""
    scale = np.float64(1.0) / np.float64(N); eps = np.float64(np.float32(1.1920928955078125e-7))
    mu1 = np.float64(0); q1 = np.float64(0); out = np.zeros(256)
    for i in range(256):
        p = np.float64(h[i]) * scale; mu1 = mu1 * q1; q1 = q1 + p; q2 = np.float64(1.0) - q1
        if not (min(q1, q2) < eps or max(q1, q2) > np.float64(1.0) - eps):
            mu1 = (mu1 + np.float64(i) * p) / q1
        out[i] = mu1
    return out


def test_one_wave_kernel_equals_the_serial_statement_and_the_oracle():
    rng = np.random.default_rng(7)
    hist = _hists(rng, 600)
    ref, _ = _run(hist, 0)
    ref4 = ref.reshape(-1)[:600 * 4].reshape(600, 4).view(np.float32)          # variant 0 packs n x 4 floats
    got, mu1 = _run(hist, 1, want_mu1=True)
    assert np.array_equal(got[:, :257], hist)                                   # result block carries the histogram and the maximum
    thr = got[:, 257:261].view(np.float32)
    assert np.array_equal(thr.view(np.int32), ref4.view(np.int32)), np.nonzero((thr != ref4).any(axis=1))[0][:10]
    for k in range(600):
        h = np.ascontiguousarray(hist[k, :256])
        assert thr[k, 2] == np.float32(O.otsu(h)) and thr[k, 3] == np.float32(O.triangle(h)), k
    for k in range(0, 600, 7):                                                  # the division chain, value by value
        want = _mu1_chain(hist[k, :256])
        assert np.array_equal(mu1[k].view(np.int64), want.view(np.int64)), (k, np.nonzero(mu1[k] != want)[0][:5])


def test_production_form_clears_the_working_histogram():
    rng = np.random.default_rng(11)
    hist = _hists(rng, 40)
    ref, _ = _run(hist, 1)
    got, _ = _run(hist, 2)
    assert np.array_equal(got[:, 257:], ref[:, 257:]) and not got[:, :257].any()
