"""DynaDetect — Python mirror of ORB_SLAM2::DynaDetect (reference include/DynaDetect.h:95-131) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr


class DynaDetect:
    """DynaDetect(imgLast, imgLastLast, fx, fy, cx, cy, depthScale); DetectDynaArea(img, imgDepth, nImg) -> (imgDyna, imgLabel)."""

    def __init__(self, imgLast: np.ndarray, imgLastLast: np.ndarray, fx: float, fy: float, cx: float, cy: float, depthScale: float,
                 device: int = 0, debug: bool = True, overlap: bool = True):
        assert imgLast.dtype == np.uint8 and imgLast.ndim == 3 and imgLast.shape[2] == 3, "CV_8UC3 BGR expected"
        self.h, self.w = imgLast.shape[:2]
        h = C.c_void_p()
        check(lib().sind_dyna_create(self.w, self.h, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(depthScale), device,
                                     C.byref(h)), "sind_dyna_create")
        self._h = h
        check(lib().sind_dyna_set_debug(self._h, 1 if debug else 0), "sind_dyna_set_debug")        # debug(): the stage images of the last frame (off = the lean drop-in path)
        check(lib().sind_dyna_set_overlap(self._h, 1 if overlap else 0), "sind_dyna_set_overlap")
        check(lib().sind_dyna_prime(self._h, ptr(np.ascontiguousarray(imgLast)), ptr(np.ascontiguousarray(imgLastLast)), self.w * 3), "sind_dyna_prime")

    def set_flow_max_levels(self, n: int):
        """build-side option of BASELINE.json config 5 ("3-level flow pyramid"): finest n levels of the DeepFlow pyramid only; 0 = all"""
        check(lib().sind_dyna_set_flow_max_levels(self._h, int(n)), "sind_dyna_set_flow_max_levels")

    def timing(self, reset: bool = True):
        """mean milliseconds per DetectDynaArea call since the last reset, by stage"""
        ms = (C.c_double * 12)(); n = check(lib().sind_dyna_timing(self._h, ms, 1 if reset else 0), "sind_dyna_timing")
        return dict(calls=n, upload=ms[0], dense_flow=ms[1], wait_depth_half=ms[2], flow_masks_and_fusion=ms[3], depth_half=ms[4], total=ms[5],
                    tail_stages=dict(flow_masks=ms[6], kmeans=ms[7], labels=ms[8], cal_occluded=ms[9], seg_and_merge=ms[10], fusion=ms[11]))

    def timing_fine(self, reset: bool = True):
        """summed milliseconds of the tail's sub-stages since the last reset (sind_dyna_timing_fine; indices in include/sind_hip.h)"""
        ms = (C.c_double * 40)(); check(lib().sind_dyna_timing_fine(self._h, ms, 1 if reset else 0), "sind_dyna_timing_fine")
        return [float(v) for v in ms]

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_dyna_destroy(self._h); self._h = None

    __del__ = close

    def DetectDynaArea(self, img: np.ndarray, imgDepth: np.ndarray, nImg: int = 0):
        img = np.ascontiguousarray(img, np.uint8); imgDepth = np.ascontiguousarray(imgDepth, np.uint16)
        dyna = np.empty((self.h, self.w), np.uint8); label = np.empty((self.h, self.w), np.uint8)
        check(lib().sind_dyna_detect(self._h, ptr(img), self.w * 3, ptr(imgDepth), self.w * 2, ptr(dyna), ptr(label), nImg), "sind_dyna_detect")
        return dyna, label

    def dilate15(self, dyna: np.ndarray) -> np.ndarray:
        """the caller-side morphologyEx(imDynaMask, DILATE, ellipse 15x15) of rgbd_tum_noros.cc:138"""
        out = np.array(dyna, np.uint8, copy=True, order="C")
        check(lib().sind_dyna_dilate15(self._h, ptr(out)), "sind_dyna_dilate15"); return out

    def debug(self):
        h, w = self.h, self.w; fw, fh = int(np.float32(0.6) * w), int(np.float32(0.6) * h)
        d = dict(flow_deep=np.zeros((2, fh, fw), np.float32), flow_refined=np.zeros((2, fh, fw), np.float32), flow_full=np.zeros((2, h, w), np.float32),
                 H=np.zeros(9), thr=np.zeros(5, np.float32), hist=np.zeros(256, np.int32), mask_low=np.zeros((h, w), np.uint8),
                 mask_high=np.zeros((h, w), np.uint8), kmeans_label=np.zeros((h, w), np.uint8), centers=np.zeros((12, 3), np.float32),
                 occ1=np.zeros((h, w), np.uint8), occ2=np.zeros((h, w), np.uint8), total_area=np.zeros((h, w), np.uint8),
                 grad_edge=np.zeros((h, w), np.uint8), plane_contours=np.zeros((h, w), np.uint8), info=np.zeros(3, np.int32))
        keys = ["flow_deep", "flow_refined", "flow_full", "H", "thr", "hist", "mask_low", "mask_high", "kmeans_label", "centers", "occ1", "occ2",
                "total_area", "grad_edge", "plane_contours", "info"]
        check(lib().sind_dyna_debug(self._h, *[ptr(d[k]) for k in keys]), "sind_dyna_debug")
        d["H"] = d["H"].reshape(3, 3); return d
