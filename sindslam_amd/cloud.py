"""Key-frame cloud generation of the mapping consumer — Python mirror of generatePointCloud(imgRGB, imgDepth, imgDepthLast, imgDynaMask,
imgDynaMaskLast, imgLabel, poseRelative, Twc) (reference octomap_pub/src/pubPointCloud.cc:471-668) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr

POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1")])


class CloudGenerator:
    def __init__(self, fx, fy, cx, cy, depth_scale, width=640, height=480, max_batch=1, device=0):
        h = C.c_void_p()
        check(lib().sind_cloud_create(C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), C.c_double(depth_scale), width, height,
                                      max_batch, device, C.byref(h)), "sind_cloud_create")
        self._h = h; self.width, self.height, self.max_batch = width, height, max_batch
        self.cap = lib().sind_cloud_max_points(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_cloud_destroy(self._h); self._h = None

    __del__ = close

    def generatePointCloud(self, imgRGB, imgDepth, imgDepthLast, imgDynaMask, imgDynaMaskLast, imgLabel, poseRelative, Twc):
        """All images [B, H, W(, 3)]; poses [B, 4, 4] float64 -> list of dict(points, occlusion, label_count, kept)"""
        B = len(imgDepth)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8); u16 = lambda a: np.ascontiguousarray(a, np.uint16); f64 = lambda a: np.ascontiguousarray(a, np.float64)
        bgr, d, dl, m, ml, lb, pr, tw = u8(imgRGB), u16(imgDepth), u16(imgDepthLast), u8(imgDynaMask), u8(imgDynaMaskLast), u8(imgLabel), f64(poseRelative), f64(Twc)
        assert bgr.shape == (B, self.height, self.width, 3) and d.shape == dl.shape == m.shape == ml.shape == lb.shape == (B, self.height, self.width)
        assert pr.shape == tw.shape == (B, 4, 4)
        pts = np.zeros((B, self.cap), POINT_DTYPE); n = np.zeros(B, np.int32)
        occ = np.zeros((B, 12), np.int32); cnt = np.zeros((B, 12), np.int32); kept = np.zeros((B, 12), np.int32)
        check(lib().sind_cloud_generate(self._h, B, ptr(bgr), ptr(d), ptr(dl), ptr(m), ptr(ml), ptr(lb), ptr(pr), ptr(tw), 0, ptr(pts), self.cap, ptr(n),
                                        ptr(occ), ptr(cnt), ptr(kept)), "sind_cloud_generate")
        return [dict(points=pts[b, :n[b]].copy(), occlusion=occ[b], label_count=cnt[b], kept=kept[b]) for b in range(B)]
