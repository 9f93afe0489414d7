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


def _read_chunked(path, n, w, h):
    raw = open(path, "rb").read(); off = 0; npx = w * h; out = {}
    for f in range(n):
        owned = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        if not owned:
            continue
        dyna = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        label = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        mask = np.frombuffer(raw, np.uint8, npx, off).reshape(h, w); off += npx
        nk = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        kps = np.frombuffer(raw, KP_DTYPE, nk, off); off += nk * KP_DTYPE.itemsize
        desc = np.frombuffer(raw, np.uint8, nk * 32, off).reshape(nk, 32); off += nk * 32
        out[f] = (dyna, label, mask, kps, desc)
    assert off == len(raw)
    return out


@pytest.mark.timeout(1200)
def test_cpp_chunked_mode_equals_the_cpp_frame_loop(tmp_path):
    """rgbd_tum_noros_shim --chunks N (sind_seq_* from C++: verified chunks, no Python, no torch.distributed) returns for every frame what the same binary's frame loop
    returns -- on one rank, and on two ranks (two processes on this card, the exchange over loopback TCP) that each write the frames they own"""
    import os
    from sindslam_amd.synth import SyntheticStream
    n = 41
    bgr, depth = SyntheticStream(seed=515).frames(0, n); h, w = bgr.shape[1:3]
    exe = cpp_shim.build(str(tmp_path / "rgbd_tum_noros_shim"))
    fin = str(tmp_path / "in.raw")
    with open(fin, "wb") as f:
        np.array([n, w, h], np.int32).tofile(f); np.ascontiguousarray(bgr).tofile(f); np.ascontiguousarray(depth).tofile(f)
    args = [repr(float(a)) if isinstance(a, float) else str(a) for a in [TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"], 1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"], 1]]
    r = subprocess.run([exe, fin, str(tmp_path / "loop.raw")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "loop.raw", "rb").read(); off = 0; npx = w * h; loop = []
    for ni in range(n):
        rec = [np.frombuffer(raw, np.uint8, npx, off + k * npx).reshape(h, w) for k in range(3)]; off += 3 * npx
        nk = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        rec.append(np.frombuffer(raw, KP_DTYPE, nk, off)); off += nk * KP_DTYPE.itemsize
        rec.append(np.frombuffer(raw, np.uint8, nk * 32, off).reshape(nk, 32)); off += nk * 32
        loop.append(rec)

    def same(got, frames):
        for f in frames:
            for a, b in zip(got[f][:3], loop[f][:3]):
                assert np.array_equal(a, b), f
            assert got[f][3].tobytes() == loop[f][3].tobytes() and np.array_equal(got[f][4], loop[f][4]), f
    # one rank, 4 chunks, a warm-up of 2 frames (short on purpose: seams mismatch and are repaired)
    r = subprocess.run([exe, fin, str(tmp_path / "c1.raw")] + args + ["--chunks", "4", "--warmup", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    print(r.stdout.strip())
    got = _read_chunked(tmp_path / "c1.raw", n, w, h); assert sorted(got) == list(range(n)); same(got, range(n))
    # two ranks x 2 chunks over TCP
    port = 36000 + os.getpid() % 20000
    ps = [subprocess.Popen([exe, fin, str(tmp_path / f"c2_{k}.raw")] + args + ["--chunks", "2", "--warmup", "1", "--rank", str(k), "--world", "2", "--port", str(port)]) for k in range(2)]
    for p in ps:
        assert p.wait(timeout=900) == 0
    g0 = _read_chunked(tmp_path / "c2_0.raw", n, w, h); g1 = _read_chunked(tmp_path / "c2_1.raw", n, w, h)
    assert sorted(list(g0) + list(g1)) == list(range(n)) and len(g1) >= 10
    same(g0, g0); same(g1, g1)
