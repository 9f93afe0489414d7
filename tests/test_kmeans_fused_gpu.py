"""GPU: the fused coarse-level k-means (k_km_level_fused: every pass of a pyramid level in one launch, one workgroup per frame, each wave summing whole runs from memory)
against the per-pass kernels and the oracle: labels, centres and everything downstream are equal bit for bit -- single frames, warm-started sequences, frames whose
depth leaves clusters empty, the batched chain of the pipeline, 1280 x 720."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd._lib import check, lib
from sindslam_amd.synth import D455, TUM3, SyntheticStream

pytestmark = pytest.mark.gpu
K3 = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])


def fused(n):
    check(lib().sind_debug_set_kmeans_fused_max(int(n)), "sind_debug_set_kmeans_fused_max")
    check(lib().sind_debug_set_kmeans_fused_min_batch(1), "sind_debug_set_kmeans_fused_min_batch")          # the tests run single frames and three streams: fuse from one frame on


@pytest.fixture(autouse=True)
def _restore():
    yield
    fused(81920); check(lib().sind_debug_set_kmeans_fused_min_batch(32))


def _run(bgr, depth, K, nf):
    from sindslam_amd.dyna import DynaDetect
    fused(nf)
    dd = DynaDetect(bgr[1], bgr[0], *K); out = []
    for f in range(2, len(bgr)):
        dy, lb = dd.DetectDynaArea(bgr[f], depth[f], f); g = dd.debug()
        out.append((dy, lb, g["kmeans_label"].copy(), g["centers"].copy()))
    dd.close()
    return out


def test_fused_levels_equal_the_per_pass_kernels_and_the_oracle():
    bgr, depth = SyntheticStream(seed=11).frames(0, 7)
    depth = depth.copy()
    depth[4][:, :] = 0                                      # a frame without valid depth: every point is (0, 0, 0), eleven clusters run empty and are repaired
    depth[5][100:300, 50:600] = 0                           # a big hole
    a = _run(bgr, depth, K3, 0); b = _run(bgr, depth, K3, 81920); c = _run(bgr, depth, K3, 400000)      # none / coarse levels / every level fused (307 200 points on level 0)
    ref = O.DynaDetect(bgr[1], bgr[0], *K3)
    for i, f in enumerate(range(2, 7)):
        rd, rl = ref.detect(bgr[f], depth[f]); r = ref.debug()
        for name, x in (("per-pass", a[i]), ("coarse fused", b[i]), ("all fused", c[i])):
            assert np.array_equal(x[2], r["kmeans_label"]) and np.array_equal(x[3].view(np.uint32), r["centers"].view(np.uint32)), (name, f)
            assert np.array_equal(x[0], rd) and np.array_equal(x[1], rl), (name, f)


def test_fused_levels_in_the_batched_pipeline_and_at_1280x720():
    from sindslam_amd.pipeline import Pipeline
    s = SyntheticStream(width=1280, height=720, intr=D455, motion_scale=0.5)
    bgr, depth = s.frames(0, 4)
    Kd = (s.fx, s.fy, s.cx, s.cy, D455["depth_factor"])
    a = _run(bgr, depth, Kd, 0); b = _run(bgr, depth, Kd, 81920)           # 14 400 and 57 600 points fused, 230 400 and 921 600 per pass
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    bgr, depth = SyntheticStream(seed=5).frames(0, 6)
    res = {}
    for nf in (0, 81920):
        fused(nf)
        p = Pipeline(3, 2, 640, 480, *K3, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1)
        sb = np.stack([bgr, bgr[:, ::-1], bgr[:, :, ::-1]]); sd = np.stack([depth, depth[:, ::-1], depth[:, :, ::-1]])
        for k in range(3):
            p.prime(k, sb[k, 1], sb[k, 0])
        got = []
        for step in range(2):
            p.process(sb[:, 2 + 2 * step:4 + 2 * step], sd[:, 2 + 2 * step:4 + 2 * step]); got.append((p.dyna.copy(), p.label.copy(), p.nkp.copy(), p.desc.copy()))
        res[nf] = got; p.close()
    for x, y in zip(res[0], res[81920]):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
