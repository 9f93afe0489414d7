"""ORBextractor — Python mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:54-88) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); call it like the reference's operator()."""

    def __init__(self, nfeatures: int = 1500, scaleFactor: float = 1.2, nlevels: int = 8, iniThFAST: int = 15, minThFAST: int = 5,
                 device: int = 0):
        self.nfeatures, self.scaleFactor, self.nlevels = nfeatures, scaleFactor, nlevels
        h = C.c_void_p()
        check(lib().sind_orb_create(nfeatures, C.c_float(scaleFactor), nlevels, iniThFAST, minThFAST, device, C.byref(h)), "sind_orb_create")
        self._h = h
        self.cap = 2 * nfeatures + 256

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_orb_destroy(self._h); self._h = None

    __del__ = close

    def reserve(self, width, height, max_batch):
        check(lib().sind_orb_reserve(self._h, width, height, max_batch), "sind_orb_reserve")

    def __call__(self, image: np.ndarray, mask: np.ndarray | None = None):
        """image: u8 [H, W]; mask: u8 [H, W] (255 = dynamic) or None -> (keypoints[KP_DTYPE], descriptors u8 [n, 32])"""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "image.type() == CV_8UC1"
        k, d = self.extract_batch(image[None], None if mask is None else mask[None])
        return k[0], d[0]

    def extract_batch(self, images: np.ndarray, masks: np.ndarray | None = None):
        images = np.ascontiguousarray(images, np.uint8); B, h, w = images.shape
        if masks is not None:
            masks = np.ascontiguousarray(masks, np.uint8)
        kps = np.zeros((B, self.cap), KP_DTYPE); desc = np.zeros((B, self.cap, 32), np.uint8); n = np.zeros(B, np.int32)
        check(lib().sind_orb_extract_batch(self._h, ptr(images), w, h, B, ptr(masks), ptr(kps), self.cap, ptr(n), ptr(desc)), "sind_orb_extract_batch")
        return [kps[b, :n[b]].copy() for b in range(B)], [desc[b, :n[b]].copy() for b in range(B)]

    # getters of the reference class
    def tables(self):
        nl = self.nlevels
        t = dict(scale=np.zeros(nl, np.float32), inv_scale=np.zeros(nl, np.float32), sigma2=np.zeros(nl, np.float32),
                 inv_sigma2=np.zeros(nl, np.float32), per_level=np.zeros(nl, np.int32), umax=np.zeros(16, np.int32))
        check(lib().sind_orb_tables(self._h, *[ptr(t[k]) for k in ["scale", "inv_scale", "sigma2", "inv_sigma2", "per_level", "umax"]]))
        return t

    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self.scaleFactor
    def GetScaleFactors(self): return self.tables()["scale"]
    def GetInverseScaleFactors(self): return self.tables()["inv_scale"]
    def GetScaleSigmaSquares(self): return self.tables()["sigma2"]
    def GetInverseScaleSigmaSquares(self): return self.tables()["inv_sigma2"]

    def image_pyramid(self, level: int, frame: int = 0):
        """mvImagePyramid[level] with its 19-px border (padded array)."""
        w = C.c_int(); h = C.c_int()
        check(lib().sind_orb_pyramid(self._h, frame, level, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value + 38, w.value + 38), np.uint8)
        check(lib().sind_orb_pyramid(self._h, frame, level, ptr(out), C.byref(w), C.byref(h)))
        return out

    def debug_fast(self, level, frame=0, cap=200000):
        xyr = np.zeros((cap, 3), np.float32); n = check(lib().sind_orb_debug_fast(self._h, frame, level, ptr(xyr), cap)); return xyr[:n].copy()

    def debug_selected(self, frame=0):
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        n = check(lib().sind_orb_debug_selected(self._h, frame, ptr(kps), self.cap, ptr(desc))); return kps[:n].copy(), desc[:n].copy()
