"""rgbd_tum_noros-shaped harness: TUM association files, OpenCV-YAML settings subset, PNG I/O and the frame loop of
the reference's Examples/RGB-D/rgbd_tum_noros.cc:37-215 around the GPU DynaDetect + ORBextractor.

    python -m sindslam_amd.harness TUM3.yaml /data/rgbd_dataset_freiburg3_walking_xyz associations.txt [--out DIR] [--max N]

What it reproduces from the reference main(): LoadImages (:217-242), the settings keys it reads (:82-86 and src/Tracking.cc:103-119),
frame 0 passed through with an all-zero mask (:100-116), DetectDynaArea from frame 1 on with frame 0 as both previous frames
(:103-107, 132-135), the 15x15 dilation (:108, 138), ORB extraction with the mask as Frame::ExtractORB2 does (src/Frame.cc:308) and the
"mean dynamic detecting time" report (:198-209).  Tracking / mapping (System::TrackRGBD) are out of scope; keypoints are returned.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

from ._lib import check, lib, ptr


# ------------------------------------------------------------------ association file (rgbd_tum_noros.cc:217-242)
def load_associations(path: str):
    """lines 'ts rgb_path ts depth_path' -> (timestamps, rgb paths, depth paths)"""
    ts, rgb, dep = [], [], []
    with open(path) as f:
        for line in f:
            s = line.split()
            if not s or s[0].startswith("#"):
                continue
            if len(s) < 4:
                raise ValueError(f"{path}: malformed association line {line!r}")
            ts.append(float(s[0])); rgb.append(s[1]); dep.append(s[3])
    if not rgb:
        raise ValueError("No images found in provided path.")
    return ts, rgb, dep


# ------------------------------------------------------------------ settings (cv::FileStorage YAML subset: 'Key.name: value')
def read_settings(path: str) -> dict:
    out = {}
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line or line.startswith("%") or ":" not in line:
                continue
            k, v = line.split(":", 1)
            v = v.strip().strip('"')
            try:
                out[k.strip()] = int(v) if v.lstrip("-").isdigit() else float(v)
            except ValueError:
                out[k.strip()] = v
    need = ["Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy", "DepthMapFactor", "ORBextractor.nFeatures", "ORBextractor.scaleFactor",
            "ORBextractor.nLevels", "ORBextractor.iniThFAST", "ORBextractor.minThFAST"]
    missing = [k for k in need if k not in out]
    if missing:
        raise KeyError(f"{path}: missing settings {missing}")
    return out


# ------------------------------------------------------------------ PNG (8-bit gray / RGB / RGBA, 16-bit gray; non-interlaced)
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def read_png(path: str) -> np.ndarray:
    """-> u8 [H, W] / [H, W, 3] in **BGR** order (what cv::imread(..., -1) hands the reference) or u16 [H, W]"""
    data = open(path, "rb").read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos = 8; idat = []; hdr = None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8]); body = data[pos + 8:pos + 8 + n]; pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 6) or (depth == 16 and ctype != 0):
        raise ValueError(f"{path}: unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 6: 4}[ctype]; bpp = ch * depth // 8; stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    out = np.empty(h * stride, np.uint8)
    check(lib().sind_png_unfilter(ptr(np.ascontiguousarray(raw)), h, stride, bpp, ptr(out)), "sind_png_unfilter")
    if depth == 16:
        return out.reshape(h, w, 2).astype(np.uint16)[..., 0] << 8 | out.reshape(h, w, 2)[..., 1]
    img = out.reshape(h, w, ch)
    if ch == 1:
        return img[..., 0].copy()
    return np.ascontiguousarray(img[..., 2::-1])        # RGB(A) -> BGR


def write_png(path: str, img: np.ndarray):
    """u8 [H, W], u8 [H, W, 3] (BGR in, RGB on disk) or u16 [H, W]; filter 0 rows"""
    if img.dtype == np.uint16:
        h, w = img.shape; depth, ctype = 16, 0; rows = img.astype(">u2").view(np.uint8).reshape(h, w * 2)
    elif img.ndim == 2:
        h, w = img.shape; depth, ctype = 8, 0; rows = np.ascontiguousarray(img, np.uint8)
    else:
        h, w, _ = img.shape; depth, ctype = 8, 2; rows = np.ascontiguousarray(img[..., ::-1], np.uint8).reshape(h, w * 3)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------ the frame loop
class GpuBackend:
    """the two drop-in classes on the MI355X (the default, and the only backend this package ships: there is no CPU fallback)"""
    def __init__(self, device: int = 0):
        self.device = device

    def make_detector(self, img_last, img_lastlast, fx, fy, cx, cy, depth_scale):
        from .dyna import DynaDetect
        return DynaDetect(img_last, img_lastlast, fx, fy, cx, cy, depth_scale, device=self.device)

    def make_extractor(self, nfeatures, scale_factor, nlevels, ini_th, min_th):
        from .orb import ORBextractor
        return ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th, device=self.device)


def run_sequence(settings_path: str, sequence_dir: str, association_path: str, out_dir: str | None = None, max_frames: int | None = None,
                 device: int = 0, verbose: bool = True, backend=None):
    """-> list of per-frame dicts {timestamp, dyna, label, mask, keypoints, descriptors}; writes masks as PNG when out_dir is given.
    backend: factory of the two classes (make_detector / make_extractor, see GpuBackend); the tests inject one built on their CPU checker to run
    the loop without a GPU (BASELINE.json configs[0], "20 frames, CPU path via rgbd_tum_noros (plumbing, no GPU)")."""
    backend = backend or GpuBackend(device)
    S = read_settings(settings_path)
    ts, rgbs, deps = load_associations(association_path)
    if len(rgbs) != len(deps):
        raise ValueError("Different number of images for rgb and depth.")
    n = len(rgbs) if max_frames is None else min(len(rgbs), max_frames)
    first = read_png(os.path.join(sequence_dir, rgbs[0]))
    dd = backend.make_detector(first, first.copy(), S["Camera.fx"], S["Camera.fy"], S["Camera.cx"], S["Camera.cy"], S["DepthMapFactor"])
    orb = backend.make_extractor(int(S["ORBextractor.nFeatures"]), float(S["ORBextractor.scaleFactor"]), int(S["ORBextractor.nLevels"]),
                                 int(S["ORBextractor.iniThFAST"]), int(S["ORBextractor.minThFAST"]))
    rgb_order = int(S.get("Camera.RGB", 0)) == 1
    h, w = first.shape[:2]
    results = []; t_dyn = []
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    for ni in range(n):
        bgr = read_png(os.path.join(sequence_dir, rgbs[ni])); depth = read_png(os.path.join(sequence_dir, deps[ni]))
        if bgr.ndim != 3 or depth.dtype != np.uint16:
            raise ValueError(f"frame {ni}: expected an 8-bit colour image and a 16-bit depth image")
        mask = np.zeros((h, w), np.uint8); dyna = mask.copy(); label = mask.copy()
        t0 = time.perf_counter()
        if ni >= 1:
            dyna, label = dd.DetectDynaArea(bgr, depth, ni)
            mask = dd.dilate15(dyna)
        t_dyn.append(time.perf_counter() - t0)
        # Tracking::GrabImageRGBD (src/Tracking.cc:246-259): RGB2GRAY on the BGR buffer when Camera.RGB is 1
        b, g, r = bgr[..., 0].astype(np.int32), bgr[..., 1].astype(np.int32), bgr[..., 2].astype(np.int32)
        gray = ((b * (4899 if rgb_order else 1868) + g * 9617 + r * (1868 if rgb_order else 4899) + 8192) >> 14).astype(np.uint8)
        kps, desc = orb(gray, mask)
        results.append(dict(timestamp=ts[ni], dyna=dyna, label=label, mask=mask, keypoints=kps, descriptors=desc))
        if out_dir:
            write_png(os.path.join(out_dir, f"dynaMask_{ni:05d}.png"), mask); write_png(os.path.join(out_dir, f"label_{ni:05d}.png"), label)
    if verbose:
        print("-------\n")
        print(f"Images in the sequence: {n}")
        print(f"mean dynamic detecting time: {sum(t_dyn) / max(len(t_dyn), 1):.6f}")
    for o in (dd, orb):
        if hasattr(o, "close"):
            o.close()
    return results


def run_sequence_chunked(settings_path: str, sequence_dir: str, association_path: str, out_dir: str | None = None, max_frames: int | None = None,
                         device: int = 0, verbose: bool = True, chunks: int = 14, frames_per_step: int = 13, warmup: int = 16):
    """The same folder, the same results (every frame equal to run_sequence's, i.e. to the reference's in-order loop), at several times the rate: the frames are
    decoded up front and go through the batched pipeline as `chunks` verified chunks (sindslam_amd.sequence.process_sequence: speculate, verify the chunk seams by
    state fingerprints, repair).  Offline use -- the whole sequence has to be there; a live camera is run_sequence / the DynaDetect class."""
    from .orb import ORBextractor
    from .seq import run_sequence as run_chunks          # the C++ driver behind sind_seq_* (csrc/host/seq.cpp)
    S = read_settings(settings_path)
    ts, rgbs, deps = load_associations(association_path)
    if len(rgbs) != len(deps):
        raise ValueError("Different number of images for rgb and depth.")
    n = len(rgbs) if max_frames is None else min(len(rgbs), max_frames)
    if n < 3:
        return run_sequence(settings_path, sequence_dir, association_path, out_dir, max_frames, device, verbose)
    t_load = time.perf_counter()
    first = read_png(os.path.join(sequence_dir, rgbs[0])); h, w = first.shape[:2]
    bgr = np.empty((n, h, w, 3), np.uint8); depth = np.empty((n, h, w), np.uint16)
    for ni in range(n):
        b = read_png(os.path.join(sequence_dir, rgbs[ni])); d = read_png(os.path.join(sequence_dir, deps[ni]))
        if b.ndim != 3 or d.dtype != np.uint16 or b.shape[:2] != (h, w) or d.shape != (h, w):
            raise ValueError(f"frame {ni}: expected an 8-bit colour image and a 16-bit depth image of {w} x {h}")
        bgr[ni] = b; depth[ni] = d
    t_load = time.perf_counter() - t_load
    intr = dict(fx=S["Camera.fx"], fy=S["Camera.fy"], cx=S["Camera.cx"], cy=S["Camera.cy"], depth_factor=S["DepthMapFactor"],
                ini_th=int(S["ORBextractor.iniThFAST"]), min_th=int(S["ORBextractor.minThFAST"]))
    rgb_order = int(S.get("Camera.RGB", 0)) == 1
    nf, sf, nl = int(S["ORBextractor.nFeatures"]), float(S["ORBextractor.scaleFactor"]), int(S["ORBextractor.nLevels"])
    st = {}; t0 = time.perf_counter()
    got = run_chunks(bgr, depth, intr, streams=max(1, min(chunks, (n - 1) // 2)), frames_per_step=frames_per_step, warmup=warmup, nfeatures=nf, scale_factor=sf, nlevels=nl,
                     orb_gray_rgb_order=1 if rgb_order else 0, device=device, stats=st)
    t_run = time.perf_counter() - t0
    # frame 0 passes through with an all-zero mask (rgbd_tum_noros.cc:100-116): its keypoints come from the extractor alone
    orb = ORBextractor(nf, sf, nl, intr["ini_th"], intr["min_th"], device=device)
    b, g, r = bgr[0][..., 0].astype(np.int32), bgr[0][..., 1].astype(np.int32), bgr[0][..., 2].astype(np.int32)
    gray0 = ((b * (4899 if rgb_order else 1868) + g * 9617 + r * (1868 if rgb_order else 4899) + 8192) >> 14).astype(np.uint8)
    zero = np.zeros((h, w), np.uint8); k0, d0 = orb(gray0, zero); orb.close()
    results = [dict(timestamp=ts[0], dyna=zero, label=zero.copy(), mask=zero.copy(), keypoints=k0, descriptors=d0)]
    for ni in range(1, n):
        results.append(dict(timestamp=ts[ni], dyna=got["dyna"][ni], label=got["label"][ni], mask=got["mask"][ni], keypoints=got["keypoints"][ni], descriptors=got["descriptors"][ni]))
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        for ni in range(n):
            write_png(os.path.join(out_dir, f"dynaMask_{ni:05d}.png"), results[ni]["mask"]); write_png(os.path.join(out_dir, f"label_{ni:05d}.png"), results[ni]["label"])
    if verbose:
        print("-------\n")
        print(f"Images in the sequence: {n}")
        print(f"decoded in {t_load:.2f} s; DynaDetect + ORB on {len(st['chunks'])} verified chunks: {t_run:.2f} s = {(n - 1) / t_run:.1f} frames/s "
              f"({st['mismatched_seams']} of {st['seams']} chunk seams repaired, {st['replay_frames'] + st['repair_frames']} frames re-run)")
        print(f"mean dynamic detecting time: {t_run / max(n - 1, 1):.6f}")
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="rgbd_tum_noros-shaped driver for the GPU DynaDetect + ORBextractor")
    ap.add_argument("settings"); ap.add_argument("sequence"); ap.add_argument("association")
    ap.add_argument("--out", default=None); ap.add_argument("--max", type=int, default=None)
    ap.add_argument("--chunks", type=int, default=0, help="offline: run the folder as this many verified chunks on the batched pipeline (same results as the frame loop, several times its rate)")
    a = ap.parse_args(argv)
    if a.chunks > 0:
        run_sequence_chunked(a.settings, a.sequence, a.association, a.out, a.max, chunks=a.chunks)
    else:
        run_sequence(a.settings, a.sequence, a.association, a.out, a.max)


if __name__ == "__main__":
    main()
