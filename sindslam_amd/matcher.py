"""ORBmatcher — Python mirror of the reference's frame-to-frame projection matcher (src/ORBmatcher.cc:1328-1470) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib


class _Config(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("bounds", C.c_float * 4),
                ("scale_factors", C.c_float * 16), ("nlevels", C.c_int), ("cap_last", C.c_int), ("cap_cur", C.c_int), ("max_batch", C.c_int), ("device", C.c_int)]


class _Pair(C.Structure):
    _fields_ = [("Tcw_cur", C.c_void_p), ("Tcw_last", C.c_void_p),
                ("n_last", C.c_int), ("x3Dw", C.c_void_p), ("last_valid", C.c_void_p), ("last_has_obs", C.c_void_p), ("last_octave", C.c_void_p),
                ("last_angle", C.c_void_p), ("last_desc", C.c_void_p),
                ("n_cur", C.c_int), ("cur_un_xy", C.c_void_p), ("cur_octave", C.c_void_p), ("cur_angle", C.c_void_p), ("cur_u_right", C.c_void_p),
                ("cur_desc", C.c_void_p), ("grid_start", C.c_void_p), ("grid_idx", C.c_void_p), ("cur_taken", C.c_void_p),
                ("match_of_cur", C.c_void_p), ("nmatches", C.c_void_p)]


class ORBmatcher:
    """ORBmatcher(nnratio, checkOri) of the reference; only SearchByProjection(CurrentFrame, LastFrame, th, bMono) is provided.
    A frame is a dict of arrays (see include/sind_hip.h, sind_match_pair)."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, fx, fy, cx, cy, bf, bounds, scale_factors, nnratio=0.6, checkOri=True, cap=4096, max_batch=1, device=0):
        cfg = _Config(fx, fy, cx, cy, bf, (C.c_float * 4)(*[float(b) for b in bounds]),
                      (C.c_float * 16)(*([float(s) for s in scale_factors] + [0.0] * (16 - len(scale_factors)))),
                      len(scale_factors), cap, cap, max_batch, device)
        self.checkOri, self.nnratio = checkOri, nnratio
        h = C.c_void_p()
        check(lib().sind_match_create(C.byref(cfg), C.byref(h)), "sind_match_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_match_destroy(self._h); self._h = None

    __del__ = close

    def SearchByProjection(self, pairs, th, bMono=False):
        """pairs: list of (Tcw_cur, Tcw_last, last, cur) -> list of (match_of_cur i32 [n_cur], nmatches)"""
        keep, arr = [], (_Pair * len(pairs))()
        f32 = lambda a: np.ascontiguousarray(a, np.float32); u8 = lambda a: np.ascontiguousarray(a, np.uint8); i32 = lambda a: np.ascontiguousarray(a, np.int32)
        outs = []
        for b, (tc, tl, last, cur) in enumerate(pairs):
            a = dict(Tcw_cur=f32(tc), Tcw_last=f32(tl), x3Dw=f32(last["x3Dw"]), last_valid=u8(last["valid"]), last_has_obs=u8(last["has_obs"]),
                     last_octave=i32(last["octave"]), last_angle=f32(last["angle"]), last_desc=u8(last["desc"]), cur_un_xy=f32(cur["un_xy"]),
                     cur_octave=i32(cur["octave"]), cur_angle=f32(cur["angle"]), cur_u_right=f32(cur["u_right"]), cur_desc=u8(cur["desc"]),
                     grid_start=i32(cur["grid_start"]), grid_idx=i32(cur["grid_idx"]))
            if cur.get("taken") is not None:
                a["cur_taken"] = u8(cur["taken"])
            nl, nc = len(a["last_valid"]), len(a["cur_octave"])
            a["match_of_cur"] = np.full(max(nc, 1), -1, np.int32); a["nmatches"] = np.zeros(1, np.int32)
            keep.append(a); arr[b].n_last = nl; arr[b].n_cur = nc
            for k, v in a.items():
                setattr(arr[b], k, v.ctypes.data if v.size else None)
            outs.append((a["match_of_cur"], a["nmatches"], nc))
        check(lib().sind_match_by_projection(self._h, arr, len(pairs), C.c_float(th), int(bMono), int(self.checkOri)), "sind_match_by_projection")
        return [(m[:nc].copy(), int(n[0])) for m, n, nc in outs]

    def last_rounds(self):
        return lib().sind_match_last_rounds(self._h)
