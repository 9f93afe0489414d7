"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (never imported by sindslam_amd/)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def lib():
    global _lib
    try:
        return _lib
    except NameError:
        pass
    if not os.path.exists(_SO):
        build()
    _lib = C.CDLL(_SO)
    _lib.orc_otsu.restype = C.c_double
    _lib.orc_triangle.restype = C.c_double
    _lib.orc_fast_atan2.restype = C.c_float
    _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    _lib.orc_dyna_create.restype = C.c_void_p
    _lib.orc_dyna_fork.restype = C.c_void_p
    _lib.orc_orb_create.restype = C.c_void_p
    _lib.orc_baseline_run.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class KP(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])


def bgr2gray(bgr, swap_rb=False):
    h, w, _ = bgr.shape; out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(np.ascontiguousarray(bgr)), w, h, int(swap_rb), _p(out)); return out


def resize_u8(src, dw, dh):
    h, w = src.shape; out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_u8(_p(np.ascontiguousarray(src)), w, h, dw, dh, _p(out)); return out


def resize_f32(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape[:2]; cn = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    lib().orc_resize_f32(_p(src), w, h, cn, dw, dh, _p(out)); return out


def gaussian_blur_u8(src, ksize=7, sigma=2.0):
    h, w = src.shape; out = np.empty_like(src)
    lib().orc_gaussian_blur_u8(_p(np.ascontiguousarray(src)), w, h, ksize, C.c_double(sigma), _p(out)); return out


def gaussian_blur3_f32(src, sigma=0.6):
    src = np.ascontiguousarray(src, np.float32); h, w = src.shape; out = np.empty_like(src)
    lib().orc_gaussian_blur3_f32(_p(src), w, h, C.c_double(sigma), _p(out)); return out


def otsu(hist):
    hist = np.ascontiguousarray(hist, np.int32); return lib().orc_otsu(_p(hist), int(hist.sum()))


def triangle(hist):
    hist = np.ascontiguousarray(hist, np.int32); return lib().orc_triangle(_p(hist))


def rng_gaussian(seed, sigma, n):
    out = np.empty(n, np.float32); lib().orc_rng_gaussian(C.c_uint64(seed), C.c_double(sigma), n, _p(out)); return out


def deepflow_levels(w, h):
    ws = np.zeros(256, np.int32); hs = np.zeros(256, np.int32)
    n = lib().orc_deepflow_levels(w, h, _p(ws), _p(hs), 256); return list(zip(ws[:n].tolist(), hs[:n].tolist()))


VARREF_INTER = ["warped", "Ix", "Iy", "Iz", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz", "A11", "A12", "A22", "b1", "b2", "wgt", "dWu", "dWv"]


def varref(i0, i1, wu, wv, fp_iters=5, sor_iters=5, alpha=20.0, delta=5.0, gamma=10.0, omega=1.6, want_inter=False):
    i0 = np.ascontiguousarray(i0, np.float32); i1 = np.ascontiguousarray(i1, np.float32)
    wu = np.array(wu, np.float32, copy=True); wv = np.array(wv, np.float32, copy=True)
    h, w = i0.shape
    inter = {k: np.empty((h, w), np.float32) for k in VARREF_INTER} if want_inter else None
    arr = (C.c_void_p * 17)(*[inter[k].ctypes.data for k in VARREF_INTER]) if want_inter else None
    lib().orc_varref(_p(i0), _p(i1), w, h, _p(wu), _p(wv), fp_iters, sor_iters, C.c_float(alpha), C.c_float(delta),
                     C.c_float(gamma), C.c_float(omega), arr)
    return (wu, wv, inter) if want_inter else (wu, wv)


def deepflow(i0, i1, max_levels=0):
    h, w = i0.shape; out = np.empty((h, w, 2), np.float32)
    if max_levels > 0:
        lib().orc_deepflow_levels_capped(_p(np.ascontiguousarray(i0)), _p(np.ascontiguousarray(i1)), w, h, int(max_levels), _p(out)); return out
    lib().orc_deepflow(_p(np.ascontiguousarray(i0)), _p(np.ascontiguousarray(i1)), w, h, _p(out)); return out


def find_homography(src, dst):
    src = np.ascontiguousarray(src, np.float32); dst = np.ascontiguousarray(dst, np.float32); H = np.zeros(9, np.float64)
    ok = lib().orc_find_homography(_p(src), _p(dst), len(src), _p(H)); return ok, H.reshape(3, 3)


def find_homography_prosac_ls(src, dst):
    """round 1's lighter estimator (PROSAC + least squares + Gauss-Newton), kept in the oracle for the a-8 sensitivity comparison"""
    src = np.ascontiguousarray(src, np.float32); dst = np.ascontiguousarray(dst, np.float32); H = np.zeros(9, np.float64)
    ok = lib().orc_find_homography_prosac_ls(_p(src), _p(dst), len(src), _p(H)); return ok, H.reshape(3, 3)


def morph(src, n, op):
    h, w = src.shape; out = np.empty_like(src)
    lib().orc_morph(_p(np.ascontiguousarray(src)), w, h, n, {"dilate": 0, "erode": 1, "open": 2, "close": 3}[op], _p(out)); return out


def find_contours(src, external_only=True):
    h, w = src.shape; pts = np.zeros((w * h * 4, 2), np.int32); lens = np.zeros(w * h, np.int32)
    n = lib().orc_find_contours(_p(np.ascontiguousarray(src)), w, h, int(external_only), _p(pts), len(pts), _p(lens), len(lens))
    out = []; o = 0
    for i in range(n):
        out.append(pts[o:o + lens[i]].copy()); o += lens[i]
    return out


def median5_f32(src):
    src = np.ascontiguousarray(src, np.float32); h, w = src.shape; out = np.empty_like(src)
    lib().orc_median5_f32(_p(src), w, h, _p(out)); return out


def dilate15(src):
    h, w = src.shape; out = np.empty_like(src); lib().orc_dilate15(_p(np.ascontiguousarray(src)), w, h, _p(out)); return out


class DynaDetect:
    def __init__(self, bgr_last, bgr_lastlast, fx, fy, cx, cy, depth_scale):
        self.h, self.w, _ = bgr_last.shape
        self.p = C.c_void_p(lib().orc_dyna_create(_p(np.ascontiguousarray(bgr_last)), _p(np.ascontiguousarray(bgr_lastlast)), self.w, self.h,
                                                  C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(depth_scale)))

    def __del__(self):
        if getattr(self, "p", None):
            lib().orc_dyna_destroy(self.p); self.p = None

    def fork(self):
        """another detector continuing from this one's inter-frame state"""
        o = DynaDetect.__new__(DynaDetect); o.h, o.w = self.h, self.w; o.p = C.c_void_p(lib().orc_dyna_fork(self.p)); return o

    def set_h_estimator(self, which=0, seed=0):
        """0: RHO's published scheme (seed 0 = its fixed seed), 2: round 1's PROSAC + least-squares estimator"""
        lib().orc_dyna_set_h_estimator(self.p, int(which), C.c_uint64(seed))

    def set_flow_max_levels(self, n):
        lib().orc_dyna_set_flow_max_levels(self.p, int(n))

    def detect(self, bgr, depth):
        dyna = np.empty((self.h, self.w), np.uint8); label = np.empty((self.h, self.w), np.uint8)
        lib().orc_dyna_detect(self.p, _p(np.ascontiguousarray(bgr)), _p(np.ascontiguousarray(depth)), _p(dyna), _p(label)); return dyna, label

    def flow_only(self, bgr):
        fw, fh = int(np.float32(0.6) * self.w), int(np.float32(0.6) * self.h)
        full = np.empty((self.h, self.w, 2), np.float32); deep = np.empty((fh, fw, 2), np.float32); ref = np.empty((fh, fw, 2), np.float32)
        lm = C.c_int(0)
        lib().orc_dyna_flow_only(self.p, _p(np.ascontiguousarray(bgr)), _p(full), _p(deep), _p(ref), C.byref(lm))
        return full, deep, ref, bool(lm.value)

    def detect_with_flow(self, bgr, depth, flow_full):
        dyna = np.empty((self.h, self.w), np.uint8); label = np.empty((self.h, self.w), np.uint8)
        lib().orc_dyna_detect_with_flow(self.p, _p(np.ascontiguousarray(bgr)), _p(np.ascontiguousarray(depth)),
                                        _p(np.ascontiguousarray(flow_full, np.float32)), _p(dyna), _p(label)); return dyna, label

    def debug(self):
        h, w = self.h, self.w
        d = dict(flow_full=np.empty((h, w, 2), np.float32), H=np.zeros(9), thr=np.zeros(5, np.float32), hist=np.zeros(256, np.int32),
                 mask_low=np.empty((h, w), np.uint8), mask_high=np.empty((h, w), np.uint8), kmeans_label=np.empty((h, w), np.uint8),
                 centers=np.zeros((12, 3), np.float32), occ1=np.empty((h, w), np.uint8), occ2=np.empty((h, w), np.uint8),
                 total_area=np.empty((h, w), np.uint8), mag_u8=np.empty((h, w), np.uint8), info=np.zeros(3, np.int32))
        lib().orc_dyna_debug(self.p, *[_p(d[k]) for k in ["flow_full", "H", "thr", "hist", "mask_low", "mask_high", "kmeans_label", "centers",
                                                          "occ1", "occ2", "total_area", "mag_u8", "info"]])
        d["H"] = d["H"].reshape(3, 3)
        for k in ("grad_edge", "plane_contours", "label_for_seg_edge"):
            d[k] = np.empty((h, w), np.uint8)
        lib().orc_dyna_debug2(self.p, _p(d["grad_edge"]), _p(d["plane_contours"]), _p(d["label_for_seg_edge"]))
        return d


class ORBextractor:
    def __init__(self, nfeatures=1500, scale=1.2, nlevels=8, ini_th=15, min_th=5):
        self.nlevels = nlevels
        self.p = C.c_void_p(lib().orc_orb_create(nfeatures, C.c_float(scale), nlevels, ini_th, min_th))

    def __del__(self):
        if getattr(self, "p", None):
            lib().orc_orb_destroy(self.p); self.p = None

    def extract(self, gray, mask=None, cap=20000):
        h, w = gray.shape; kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_orb_extract(self.p, _p(np.ascontiguousarray(gray)), w, h, _p(None if mask is None else np.ascontiguousarray(mask)), _p(kps), cap, _p(desc))
        return kps[:n].copy(), desc[:n].copy()

    def level_size(self, level):
        w = C.c_int(); h = C.c_int(); lib().orc_orb_level_size(self.p, level, C.byref(w), C.byref(h)); return w.value, h.value

    def level_padded(self, level):
        w, h = self.level_size(level); out = np.empty((h + 38, w + 38), np.uint8); lib().orc_orb_level_padded(self.p, level, _p(out)); return out

    def fast_keypoints(self, level, cap=100000):
        kps = np.zeros(cap, KP_DTYPE); n = lib().orc_orb_fast_keypoints(self.p, level, _p(kps), cap); return kps[:n].copy()

    def selected(self, level, cap=20000):
        kps = np.zeros(cap, KP_DTYPE); n = lib().orc_orb_selected(self.p, level, _p(kps), cap); return kps[:n].copy()

    def tables(self):
        n = self.nlevels
        t = dict(scale=np.zeros(n, np.float32), inv_scale=np.zeros(n, np.float32), sigma2=np.zeros(n, np.float32), inv_sigma2=np.zeros(n, np.float32),
                 per_level=np.zeros(n, np.int32), umax=np.zeros(16, np.int32))
        lib().orc_orb_tables(self.p, *[_p(t[k]) for k in ["scale", "inv_scale", "sigma2", "inv_sigma2", "per_level", "umax"]]); return t


STAGES = ["kmeans", "depth_edge", "seg_and_merge", "flow_refine_masks", "fusion", "orb"]


def baseline_run(bgr, depth, intr, nfeatures=1500, scale=1.2, nlevels=8, orb_gray_rgb_order=1, want_outputs=False, kp_cap=4096, warmup_pairs=0, flow_max_levels=0):
    """frames 0,1 prime the detector; the first `warmup_pairs` pairs are not timed.  Returns (seconds, stage seconds[, dyna masks (n-2,h,w),
    keypoints per pair]); stage seconds = [flow, tail, orb] + the reference's own breakdown (STAGES: DynaDetect.cc:1421,1499,1518,1161,1644 + ORB)"""
    n, h, w, _ = bgr.shape; st = np.zeros(9)
    dyna = np.zeros((n - 2, h, w), np.uint8) if want_outputs else None
    nkp = np.zeros(n - 2, np.int32) if want_outputs else None
    kps = np.zeros((n - 2, kp_cap), KP_DTYPE) if want_outputs else None
    t = lib().orc_baseline_run(_p(np.ascontiguousarray(bgr)), _p(np.ascontiguousarray(depth)), n, w, h, C.c_float(intr["fx"]), C.c_float(intr["fy"]),
                               C.c_float(intr["cx"]), C.c_float(intr["cy"]), C.c_float(intr["depth_factor"]), nfeatures, C.c_float(scale), nlevels,
                               intr["ini_th"], intr["min_th"], _p(st), int(orb_gray_rgb_order), _p(dyna), _p(nkp), _p(kps), kp_cap, int(warmup_pairs), int(flow_max_levels))
    if want_outputs:
        return t, st, dyna, [kps[i, :nkp[i]] for i in range(n - 2)]
    return t, st


def frame_post_orb(calib11, kx, ky, depth):
    """calib11 = fx fy cx cy k1 k2 p1 p2 k3 bf depthMapFactor -> dict of the Frame members (oracle/frame.hpp)"""
    c = np.asarray(calib11, np.float32); kx = np.ascontiguousarray(kx, np.float32); ky = np.ascontiguousarray(ky, np.float32)
    depth = np.ascontiguousarray(depth, np.uint16); h, w = depth.shape; n = len(kx)
    un = np.zeros((max(n, 1), 2), np.float32); ur = np.zeros(max(n, 1), np.float32); dep = np.zeros(max(n, 1), np.float32)
    cell = np.zeros(max(n, 1), np.int32); gs = np.zeros(64 * 48 + 1, np.int32); gi = np.zeros(max(n, 1), np.int32); b = np.zeros(4, np.float32)
    m = lib().orc_frame_post_orb(_p(c), _p(kx), _p(ky), n, _p(depth), w, h, _p(un), _p(ur), _p(dep), _p(cell), _p(gs), _p(gi), _p(b))
    return dict(keys_un=un[:n], u_right=ur[:n], depth=dep[:n], cell=cell[:n], grid_start=gs, grid_idx=gi[:m], bounds=b)


def search_by_projection(cam10, scale, Tcw_cur, Tcw_last, last, cur, th, mono=False, check_orientation=True):
    """last: dict(x3Dw, valid, has_obs, octave, angle, desc); cur: dict(un_xy, octave, angle, u_right, desc, grid_start, grid_idx, taken|None)
    -> (match_of_cur, nmatches)   (oracle/matcher.hpp)"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32); u8 = lambda a: np.ascontiguousarray(a, np.uint8); i32 = lambda a: np.ascontiguousarray(a, np.int32)
    cam = f32(cam10); sc = f32(scale); tc = f32(Tcw_cur); tl = f32(Tcw_last)
    L = [f32(last["x3Dw"]), u8(last["valid"]), u8(last["has_obs"]), i32(last["octave"]), f32(last["angle"]), u8(last["desc"])]
    Cc = [f32(cur["un_xy"]), i32(cur["octave"]), f32(cur["angle"]), f32(cur["u_right"]), u8(cur["desc"]), i32(cur["grid_start"]), i32(cur["grid_idx"])]
    tk = None if cur.get("taken") is None else u8(cur["taken"])
    nl, nc = len(L[1]), len(Cc[1]); out = np.full(max(nc, 1), -1, np.int32)
    n = lib().orc_search_by_projection(_p(cam), _p(sc), len(sc), _p(tc), _p(tl), nl, *[_p(a) for a in L], nc, *[_p(a) for a in Cc], _p(tk), C.c_float(th),
                                       int(mono), int(check_orientation), _p(out))
    return out[:nc], n


CLOUD_POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1")])


def generate_point_cloud(cam5, bgr, depth, depth_last, dyna, dyna_last, label, pose_relative, Twc):
    """oracle/cloud.hpp -> dict(points, occlusion, label_count, kept)"""
    h, w = depth.shape; cam = np.asarray(cam5, np.float64)
    a = [np.ascontiguousarray(bgr, np.uint8), np.ascontiguousarray(depth, np.uint16), np.ascontiguousarray(depth_last, np.uint16),
         np.ascontiguousarray(dyna, np.uint8), np.ascontiguousarray(dyna_last, np.uint8), np.ascontiguousarray(label, np.uint8)]
    pr = np.ascontiguousarray(pose_relative, np.float64); tw = np.ascontiguousarray(Twc, np.float64)
    cap = ((w + 1) // 2) * ((h + 1) // 2); pts = np.zeros(cap, CLOUD_POINT_DTYPE)
    occ = np.zeros(12, np.float64); cnt = np.zeros(12, np.int32); kept = np.zeros(12, np.uint8)
    n = lib().orc_generate_point_cloud(_p(cam), *[_p(x) for x in a], w, h, _p(pr), _p(tw), _p(pts), cap, _p(occ), _p(cnt), _p(kept))
    return dict(points=pts[:n].copy(), occlusion=occ.astype(np.int32), label_count=cnt, kept=kept.astype(np.int32))


def sequence_run(bgr, depth, intr, threads=8, swap_rb_orb=True, nfeatures=1500, want_orb=True):
    """The reference's sequential frame loop (rgbd_tum_noros.cc:110-170) on the oracle over a whole sequence: frame 0 primes twice, frames 1.. go through
    DetectDynaArea -> 15x15 dilate -> ORBextractor.  The dense flow of a frame is state free (frames n, n-1, n-2 only), so it is computed for all frames on
    `threads` host threads first (one primed scalar detector per frame; ctypes calls release the GIL) and the stateful part then runs strictly in order with
    detect_with_flow -- same results as detect() frame by frame (tests/test_oracle_sequence_cpu.py), in a fraction of the time.
    Returns lists indexed by frame (entry 0 is None): dyna, label, mask, keypoints, descriptors."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(bgr); K = (intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"])
    lib()

    def flow_of(f):
        d = DynaDetect(bgr[f - 1], bgr[max(f - 2, 0)], *K); return d.flow_only(bgr[f])[0]
    with ThreadPoolExecutor(max(1, threads)) as ex:
        flows = [None] + list(ex.map(flow_of, range(1, n)))
    det = DynaDetect(bgr[0], bgr[0].copy(), *K)
    orb = ORBextractor(nfeatures, 1.2, 8, intr["ini_th"], intr["min_th"]) if want_orb else None
    out = dict(dyna=[None], label=[None], mask=[None], keypoints=[None], descriptors=[None])
    for f in range(1, n):
        dy, lb = det.detect_with_flow(bgr[f], depth[f], flows[f]); flows[f] = None
        mk = dilate15(dy)
        out["dyna"].append(dy); out["label"].append(lb); out["mask"].append(mk)
        if want_orb:
            k, dsc = orb.extract(bgr2gray(bgr[f], swap_rb=swap_rb_orb), mk)
            out["keypoints"].append(k); out["descriptors"].append(dsc)
        else:
            out["keypoints"].append(None); out["descriptors"].append(None)
    return out
