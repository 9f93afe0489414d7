"""Frame post-ORB steps — Python mirror of what ORB_SLAM2::Frame's RGB-D constructor does with the extractor's output
(reference src/Frame.cc:143-170): UndistortKeyPoints, ComputeStereoFromRGBD, ComputeImageBounds, AssignFeaturesToGrid."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import check, lib, ptr
from .orb import KP_DTYPE

FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64           # reference include/Frame.h:37-38


class _Calib(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf", "depth_map_factor")]


@dataclass
class FrameFeatures:
    """Per-frame members the reference Frame fills: mvKeysUn (xy), mvuRight, mvDepth, mGrid."""
    keys_un: np.ndarray      # f32 [N, 2]
    u_right: np.ndarray      # f32 [N]
    depth: np.ndarray        # f32 [N]
    cell: np.ndarray         # i32 [N]  x * 48 + y or -1
    grid_start: np.ndarray   # i32 [64*48 + 1]
    grid_idx: np.ndarray     # i32 [entries]

    def grid(self, x: int, y: int) -> np.ndarray:
        """mGrid[x][y]"""
        c = x * FRAME_GRID_ROWS + y
        return self.grid_idx[self.grid_start[c]:self.grid_start[c + 1]]


class FramePostORB:
    def __init__(self, width, height, fx, fy, cx, cy, bf, depth_map_factor, dist=(0.0, 0.0, 0.0, 0.0, 0.0), max_batch=1, cap=4096, device=0):
        d = list(dist) + [0.0] * (5 - len(dist))
        self.calib = _Calib(fx, fy, cx, cy, d[0], d[1], d[2], d[3], d[4], bf, depth_map_factor)
        self.width, self.height, self.max_batch, self.cap = width, height, max_batch, cap
        h = C.c_void_p()
        check(lib().sind_frame_create(C.byref(self.calib), width, height, max_batch, cap, device, C.byref(h)), "sind_frame_create")
        self._h = h
        self.bounds = None       # mnMinX, mnMaxX, mnMinY, mnMaxY after the first call

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_frame_destroy(self._h); self._h = None

    __del__ = close

    def __call__(self, keypoints, depth):
        """keypoints: list of B KP_DTYPE arrays; depth: u16 [B, H, W] numpy array, or an int device pointer to that layout."""
        B, cap = len(keypoints), self.cap
        kps = np.zeros((B, cap), KP_DTYPE); n = np.array([len(k) for k in keypoints], np.int32)
        for b, k in enumerate(keypoints):
            if len(k) > cap:
                check(-5, f"{len(k)} keypoints exceed cap {cap}")
            kps[b, :len(k)] = k
        on_dev = isinstance(depth, int)
        if not on_dev:
            depth = np.ascontiguousarray(depth, np.uint16)
            assert depth.shape == (B, self.height, self.width), "imDepth shape"
        un = np.zeros((B, cap, 2), np.float32); ur = np.zeros((B, cap), np.float32); dep = np.zeros((B, cap), np.float32)
        cell = np.zeros((B, cap), np.int32); gs = np.zeros((B, FRAME_GRID_ROWS * FRAME_GRID_COLS + 1), np.int32); gi = np.zeros((B, cap), np.int32)
        bounds = np.zeros(4, np.float32)
        check(lib().sind_frame_post_orb(self._h, ptr(kps), ptr(n), B, ptr(depth), int(on_dev), ptr(un), ptr(ur), ptr(dep), ptr(cell), ptr(gs),
                                        ptr(gi), ptr(bounds)), "sind_frame_post_orb")
        self.bounds = bounds
        return [FrameFeatures(un[b, :n[b]].copy(), ur[b, :n[b]].copy(), dep[b, :n[b]].copy(), cell[b, :n[b]].copy(), gs[b].copy(),
                              gi[b, :gs[b, -1]].copy()) for b in range(B)]
