"""GPU: the wave-parallel EXACT sequential FP32 sum behind the k-means centres (k_km_seqsum, csrc/depth_kernels.hip) against the plain
left-to-right float32 accumulation of cv::kmeans (kmeans.cpp "compute centers"; numpy's add.accumulate is that loop) -- bit for bit, on data
chosen to hit every premise of the window arithmetic: binade crossings in both directions, exact ties, exact powers of two, zeros, huge
dynamic range, runs longer than the LDS ring, lengths that are not a multiple of anything."""
import ctypes as C

import numpy as np
import pytest

from sindslam_amd._lib import check, lib, ptr

pytestmark = pytest.mark.gpu


def seq(x):
    x = np.asarray(x, np.float32)
    return np.float32(0) if x.size == 0 else np.add.accumulate(x, dtype=np.float32)[-1]


def gpu(x):
    x = np.ascontiguousarray(x, np.float32); out = C.c_float(0)
    check(lib().sind_debug_seqsum(ptr(x) if x.size else ptr(np.zeros(1, np.float32)), int(x.size), 0, C.byref(out)), "sind_debug_seqsum")
    return np.float32(out.value)


def cases():
    rng = np.random.default_rng(2024)
    yield "empty", np.zeros(0, np.float32)
    yield "one", np.array([3.25], np.float32)
    yield "zeros", np.zeros(1000, np.float32)
    yield "leading zeros then data", np.concatenate([np.zeros(777, np.float32), rng.uniform(0, 9, 5000).astype(np.float32)])
    yield "depth-like monotone (92k)", (1.5 * rng.uniform(0.5, 6.0, 92_001)).astype(np.float32)
    yield "x-coordinates around zero (binade crossings both ways)", rng.normal(0, 1.5, 70_003).astype(np.float32)
    yield "row-wise sign flips", np.tile(np.concatenate([-rng.uniform(0, 2, 300), rng.uniform(0, 2, 300)]).astype(np.float32), 60)
    yield "interleaved zeros (hole blocks)", (rng.uniform(0.1, 8, 40_000) * (rng.random(40_000) > 0.3)).astype(np.float32)
    yield "ones through 2^k", np.ones(20_000, np.float32)
    yield "stall at 2^24 (1 is half an ulp: ties to even)", np.concatenate([[np.float32(2 ** 24 - 6)], np.ones(40, np.float32)]).astype(np.float32)
    yield "exact ties, both parities", np.concatenate([[np.float32(2 ** 23)], np.tile(np.array([0.5, 0.5, 1.5, 2.5, 0.5], np.float32), 400)]).astype(np.float32)
    yield "ties with negative accumulator", np.concatenate([[np.float32(-(2 ** 23) - 3)], np.tile(np.array([-0.5, 0.5, -1.5, 0.5], np.float32), 300)]).astype(np.float32)
    yield "down through powers of two", np.concatenate([[np.float32(4096.0)], -np.full(8000, 0.7, np.float32)]).astype(np.float32)
    yield "exactly back to a power of two and below", np.array([8.0, -0.25, -0.25, 0.5, -4.0, -0.0000004, 1e-8, -3.9999995], np.float32)
    yield "cancellation to zero and on", np.array([5.5, -5.5, 0.0, 0.0, 1e-20, 3.0, -3.0, -0.0, 7.0], np.float32)
    yield "huge dynamic range", (rng.uniform(-1, 1, 30_000) * 10.0 ** rng.integers(-30, 30, 30_000)).astype(np.float32)
    yield "tiny then big", np.concatenate([np.full(3000, 1e-30, np.float32), np.full(3000, 1e20, np.float32), np.full(10, -1e20, np.float32)])
    yield "longer than the ring (3 chunks + 3)", rng.uniform(-3, 9, 4 * 8192 + 8192 + 3).astype(np.float32)
    yield "denormals", np.full(5000, 1e-41, np.float32)
    for n in (1, 2, 3, 4, 5, 255, 256, 257, 8191, 8192, 8193, 16385):
        yield f"length {n}", rng.uniform(-2, 5, n).astype(np.float32)


@pytest.mark.parametrize("name,x", list(cases()), ids=[c[0] for c in cases()])
def test_sequential_sum_bit_exact(name, x):
    a, b = gpu(x), seq(x)
    assert a.tobytes() == b.tobytes(), (name, float(a), float(b))


def test_real_point_runs(frames):
    """the actual k-means inputs: coordinates of the back-projected depth pixels of a frame, grouped like the 3 x 4 grid labels"""
    from sindslam_amd.synth import TUM3
    _, depth = frames
    d = depth[2].astype(np.float32) / TUM3["depth_factor"]; v, u = np.mgrid[0:480, 0:640]
    z = np.where((d > 0) & (d < 6), d, 0).astype(np.float32)
    X = ((u - TUM3["cx"]) * z / TUM3["fx"]).astype(np.float32); Y = ((v - TUM3["cy"]) * z / TUM3["fy"]).astype(np.float32); Z = (z * 1.5).astype(np.float32)
    lab = (v // 160) * 4 + (u // 160)
    for k in range(12):
        for P in (X, Y, Z):
            run = P[lab == k]
            assert gpu(run).tobytes() == seq(run).tobytes(), k
