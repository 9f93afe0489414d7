"""GPU: the reference's frame loop written in C++ against the drop-in classes (examples/rgbd_tum_noros_shim.cpp, built with g++)
returns exactly what the Python mirror of the same C ABI returns, frame by frame."""
import subprocess

import numpy as np
import pytest

import cpp_shim
from sindslam_amd.orb import KP_DTYPE
from sindslam_amd.synth import TUM3

pytestmark = pytest.mark.gpu


def test_cpp_frame_loop_equals_python_mirror(frames, tmp_path):
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    bgr, depth = frames
    n, h, w, _ = bgr.shape; n = min(n, 5)
    exe = cpp_shim.build(str(tmp_path / "rgbd_tum_noros_shim"))
    fin, fout = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    with open(fin, "wb") as f:
        np.array([n, w, h], np.int32).tofile(f); np.ascontiguousarray(bgr[:n]).tofile(f); np.ascontiguousarray(depth[:n]).tofile(f)
    args = [TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"], 1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"], 1]
    r = subprocess.run([exe, fin, fout] + [repr(float(a)) if isinstance(a, float) else str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"Images in the sequence: {n}, scale factors 8" in r.stdout
    raw = open(fout, "rb").read(); off = 0; npx = w * h
    dd = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    orb = ORBextractor(1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"])
    some_dynamic = False
    for ni in range(n):
        dyna = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        label = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        mask = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        nk = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        kps = np.frombuffer(raw, KP_DTYPE, nk, off); off += nk * KP_DTYPE.itemsize
        desc = np.frombuffer(raw, np.uint8, nk * 32, off).reshape(nk, 32); off += nk * 32
        rd = np.zeros((h, w), np.uint8); rl = rd.copy(); rm = rd.copy()
        if ni >= 1:
            rd, rl = dd.DetectDynaArea(bgr[ni], depth[ni], ni); rm = dd.dilate15(rd)
        b, g, rr = bgr[ni][..., 0].astype(np.int32), bgr[ni][..., 1].astype(np.int32), bgr[ni][..., 2].astype(np.int32)
        gray = ((b * 4899 + g * 9617 + rr * 1868 + 8192) >> 14).astype(np.uint8)
        rk, rdesc = orb(gray, rm)
        assert np.array_equal(dyna, rd) and np.array_equal(label, rl) and np.array_equal(mask, rm), ni
        assert nk == len(rk) and kps.tobytes() == rk.tobytes() and np.array_equal(desc, rdesc), ni
        some_dynamic |= bool((dyna == 255).any())
    assert off == len(raw) and some_dynamic
    dd.close(); orb.close()


def test_opencv_overloads_and_public_pyramid(frames, tmp_path):
    """the -DSIND_WITH_OPENCV branch (built against the test-only <opencv2/core.hpp>) run end to end: DetectDynaArea on cv::Mat, the extractor on
    cv::InputArray / cv::Mat() / OutputArray, and mvImagePyramid (padded ROI levels) against the Python mirror of the same C ABI"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    bgr, depth = frames
    n, h, w, _ = bgr.shape; n = min(n, 3)
    exe = cpp_shim.build_boundary(str(tmp_path / "boundary_callsites"))
    fin, fout = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    with open(fin, "wb") as f:
        np.array([n, w, h], np.int32).tofile(f); np.ascontiguousarray(bgr[:n]).tofile(f); np.ascontiguousarray(depth[:n]).tofile(f)
    r = subprocess.run([exe, fin, fout] + [repr(float(TUM3[k])) for k in ("fx", "fy", "cx", "cy", "depth_factor")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr)
    raw = open(fout, "rb").read(); off = 0; npx = w * h
    dd = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    orb = ORBextractor(1500, 1.2, 8, 15, 5)
    rd = np.zeros((h, w), np.uint8); rl = rd.copy()
    for ni in range(n):
        dyna = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        label = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        nk = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        kps = np.frombuffer(raw, KP_DTYPE, nk, off); off += nk * KP_DTYPE.itemsize
        desc = np.frombuffer(raw, np.uint8, nk * 32, off).reshape(nk, 32); off += nk * 32
        if ni >= 1:
            rd, rl = dd.DetectDynaArea(bgr[ni], depth[ni], ni)
        b, g, rr = bgr[ni][..., 0].astype(np.int32), bgr[ni][..., 1].astype(np.int32), bgr[ni][..., 2].astype(np.int32)
        gray = ((b * 4899 + g * 9617 + rr * 1868 + 8192) >> 14).astype(np.uint8)
        rk, rdesc = orb(gray, rd if ni >= 1 else None)                   # the undilated mask, as the binary passes it
        assert np.array_equal(dyna, rd) and np.array_equal(label, rl), ni
        assert nk == len(rk) and kps.tobytes() == rk.tobytes() and np.array_equal(desc, rdesc), ni
    nl = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
    assert nl == 8
    for l in range(nl):
        lw, lh = np.frombuffer(raw, np.int32, 2, off); off += 8
        lvl = np.frombuffer(raw, np.uint8, lw * lh, off).reshape(lh, lw); off += lw * lh
        pad = orb.image_pyramid(l)
        assert pad.shape == (lh + 38, lw + 38) and np.array_equal(lvl, pad[19:19 + lh, 19:19 + lw]), l
        if l == 1:
            lvl1 = lvl
    patch = np.frombuffer(raw, np.uint8, 121, off).reshape(11, 11); off += 121
    assert np.array_equal(patch, lvl1[35:46, 55:66]) and off == len(raw)
    dd.close(); orb.close()
