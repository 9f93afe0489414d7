"""GPU: the region grow of the PEAC plane refinement as a level-synchronous ordered BFS (k_peac_grow) against the host statement of the reference's FIFO
(AHCPlaneFitter.hpp:546-601 floodFill): the membership map of every pixel and the set of plane pairs that met must be identical."""
import ctypes as C

import numpy as np
import pytest

from sindslam_amd._lib import check, lib, ptr
from sindslam_amd.synth import D455, TUM3, SyntheticStream

pytestmark = pytest.mark.gpu
PP = 127 * 127


def _grow(depth, intr):
    n, h, w = depth.shape
    mg = np.zeros((n, h * w), np.int8); mh = np.zeros((n, h * w), np.int8); pg = np.zeros((n, PP), np.uint8); ph = np.zeros((n, PP), np.uint8)
    st = np.zeros((n, 4), np.int32)
    check(lib().sind_debug_peac_grow(ptr(np.ascontiguousarray(depth, np.uint16)), n, w, h, C.c_float(intr["fx"]), C.c_float(intr["fy"]), C.c_float(intr["cx"]), C.c_float(intr["cy"]),
                                     C.c_float(intr["depth_factor"]), 0, ptr(mg), ptr(mh), ptr(pg), ptr(ph), ptr(st)), "sind_debug_peac_grow")
    return mg, mh, pg, ph, st


def _same(mg, mh, pg, ph, st):
    for k in range(len(st)):
        assert st[k, 0] == 0, ("kernel status", k, st[k])
        npl = st[k, 3]
        assert np.array_equal(mg[k], mh[k]), (k, int((mg[k] != mh[k]).sum()), st[k])
        assert np.array_equal(pg[k, :npl * npl], ph[k, :npl * npl]), (k, "pairs")


def test_grow_equals_the_fifo_on_stream_frames(stream):
    _, depth = stream.frames(0, 12)
    for rep in range(4):                          # the kernel's phases race with each other if a barrier or an ownership rule is wrong: such bugs show up in SOME launches
        mg, mh, pg, ph, st = _grow(depth, TUM3)
        _same(mg, mh, pg, ph, st)
    assert (st[:, 3] >= 3).all() and (st[:, 1] > 50).all() and (st[:, 2] > 50_000).all()           # planes found, a deep traversal, ~10^5 seeds per frame
    assert (mg >= 0).mean() > 0.3


def test_one_and_two_frame_launches_take_the_1024_thread_instance(stream):
    """launches of one or two frames (the drop-in path's call) run the kernel with 1024 threads per workgroup: same membership, same pairs"""
    _, depth = stream.frames(0, 4)
    for k in range(4):
        _same(*_grow(depth[k:k + 1], TUM3))
    _same(*_grow(depth[1:3], TUM3))
    sc = 2.0; intr = dict(D455, fx=D455["fx"] * sc, fy=D455["fy"] * sc, cx=D455["cx"] * sc, cy=D455["cy"] * sc)
    _, d720 = SyntheticStream(1280, 720, 4242, D455).frames(0, 2)
    _same(*_grow(d720[:1], intr)); _same(*_grow(d720[:2], intr))


def test_grow_on_a_second_scene_and_a_mirrored_one():
    s = SyntheticStream(seed=777)
    _, depth = s.frames(3, 6)
    d2 = np.concatenate([depth, depth[:, :, ::-1], depth[:, ::-1]])
    _same(*_grow(d2, TUM3))


def test_grow_at_1280x720():
    sc = 2.0; intr = dict(D455, fx=D455["fx"] * sc, fy=D455["fy"] * sc, cx=D455["cx"] * sc, cy=D455["cy"] * sc)
    s = SyntheticStream(1280, 720, 4242, D455)
    _, depth = s.frames(0, 3)
    mg, mh, pg, ph, st = _grow(depth, intr)
    _same(mg, mh, pg, ph, st)


def test_degenerate_frames():
    h, w = 480, 640
    flat = np.full((1, h, w), 7500, np.uint16)                       # one fronto-parallel plane: every block eroded, nothing to grow
    holes = flat.copy(); holes[0, ::7, ::5] = 0                       # invalid pixels sprinkled over it
    empty = np.zeros((1, h, w), np.uint16)                            # no depth at all: no planes
    rng = np.random.default_rng(5); noise = rng.integers(500, 30000, (1, h, w)).astype(np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    two = np.where(xx < 300, 5000 + 6 * xx, 12000 - 3 * yy).astype(np.uint16)[None]      # two slanted planes meeting at a depth step
    wedge = (6000 + 4 * np.abs(xx - 320) + 2 * yy).astype(np.uint16)[None]                # two planes meeting at a ridge: they DO meet during the grow
    d = np.concatenate([flat, holes, empty, noise, two, wedge])
    mg, mh, pg, ph, st = _grow(d, TUM3)
    _same(mg, mh, pg, ph, st)
    assert st[2, 3] == 0 and st[0, 3] == 1


def test_more_planes_than_the_kernel_holds_go_to_the_host():
    """130 small planes (a checkerboard of 48 x 48 facets tilted +-30 degrees in x and y -- three blocks a side, the smallest the fitter's two-apart
    linking rule joins): beyond the kernel's 127, the frame is flagged `skipped` and the host statement of the
    FIFO grows it -- end to end the detector still equals the oracle on such a frame"""
    import oracle_lib as O
    from sindslam_amd.dyna import DynaDetect
    h, w = 480, 640
    yy, xx = np.mgrid[0:h, 0:w]
    ti, tj = yy // 48, xx // 48
    sx = np.where(tj % 2 == 0, 1, -1); sy = np.where(ti % 2 == 0, 1, -1)
    depth = (7500 + sx * 8 * (xx - tj * 48 - 24) + sy * 8 * (yy - ti * 48 - 24)).astype(np.uint16)
    mg, mh, pg, ph, st = _grow(depth[None], TUM3)
    assert st[0, 3] > 127 and st[0, 0] == 4, st
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 255, (3, h, w, 3), dtype=np.uint8)
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    gpu = DynaDetect(bgr[1], bgr[0], *K); ref = O.DynaDetect(bgr[1], bgr[0], *K)
    gd, gl = gpu.DetectDynaArea(bgr[2], depth, 2); rd, rl = ref.detect(bgr[2], depth)
    g, r = gpu.debug(), ref.debug()
    assert np.array_equal(g["occ1"], r["occ1"]) and np.array_equal(g["occ2"], r["occ2"]) and np.array_equal(gd, rd) and np.array_equal(gl, rl)
