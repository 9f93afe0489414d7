"""CPU: the rgbd_tum_noros-shaped harness pieces that need no GPU (association parser, settings reader, PNG codec)."""
import os
import struct
import zlib

import numpy as np
import pytest

from sindslam_amd import harness as Hn


def test_association_parser(tmp_path):
    p = tmp_path / "assoc.txt"
    p.write_text("1305031102.175304 rgb/a.png 1305031102.160407 depth/a.png\n\n1305031102.211214 rgb/b.png 1305031102.226738 depth/b.png\n")
    ts, rgb, dep = Hn.load_associations(str(p))
    assert ts == [1305031102.175304, 1305031102.211214] and rgb == ["rgb/a.png", "rgb/b.png"] and dep == ["depth/a.png", "depth/b.png"]
    (tmp_path / "empty.txt").write_text("\n")
    with pytest.raises(ValueError, match="No images found"):
        Hn.load_associations(str(tmp_path / "empty.txt"))


def test_settings_reader(tmp_path):
    p = tmp_path / "TUM3.yaml"
    p.write_text("%YAML:1.0\n# comment\nCamera.fx: 535.4\nCamera.fy: 539.2\nCamera.cx: 320.1\nCamera.cy: 247.6\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n"
                 "ORBextractor.nFeatures: 1500\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 15   # inline\nORBextractor.minThFAST: 5\n")
    s = Hn.read_settings(str(p))
    assert s["Camera.fx"] == 535.4 and s["DepthMapFactor"] == 5000.0 and s["ORBextractor.nFeatures"] == 1500 and s["ORBextractor.iniThFAST"] == 15 and s["Camera.RGB"] == 1
    (tmp_path / "bad.yaml").write_text("%YAML:1.0\nCamera.fx: 1.0\n")
    with pytest.raises(KeyError):
        Hn.read_settings(str(tmp_path / "bad.yaml"))


def _filter_rows(rows, bpp, ftypes):
    """reference PNG filtering (spec section 9) used to build test files with every filter type"""
    h, stride = rows.shape; out = bytearray()
    for y in range(h):
        ft = ftypes[y % len(ftypes)]; out.append(ft); cur = rows[y].astype(np.int32); up = rows[y - 1].astype(np.int32) if y else np.zeros(stride, np.int32)
        for x in range(stride):
            a = cur[x - bpp] if x >= bpp else 0; b = up[x]; c = up[x - bpp] if (y and x >= bpp) else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c); pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((int(cur[x]) - pred) & 255)
    return bytes(out)


def _png(w, h, depth, ctype, raw):
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_png_round_trip_and_all_filters(tmp_path):
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (20, 31, 3), dtype=np.uint8); d16 = rng.integers(0, 65536, (20, 31), dtype=np.uint16); g8 = rng.integers(0, 256, (9, 7), dtype=np.uint8)
    for name, img in (("c.png", bgr), ("d.png", d16), ("g.png", g8)):
        Hn.write_png(str(tmp_path / name), img)
        back = Hn.read_png(str(tmp_path / name))
        assert back.dtype == img.dtype and np.array_equal(back, img)
    rgb_rows = bgr[..., ::-1].reshape(20, 31 * 3)
    (tmp_path / "f.png").write_bytes(_png(31, 20, 8, 2, _filter_rows(rgb_rows, 3, [0, 1, 2, 3, 4])))
    assert np.array_equal(Hn.read_png(str(tmp_path / "f.png")), bgr)
    rows16 = d16.astype(">u2").view(np.uint8).reshape(20, 62)
    (tmp_path / "f16.png").write_bytes(_png(31, 20, 16, 0, _filter_rows(rows16, 2, [4, 3, 1, 2])))
    assert np.array_equal(Hn.read_png(str(tmp_path / "f16.png")), d16)
    (tmp_path / "bad.png").write_bytes(b"not a png")
    with pytest.raises(ValueError):
        Hn.read_png(str(tmp_path / "bad.png"))


class _OracleBackend:
    """the harness' two classes built on the CPU checker (test infrastructure; the product has no CPU path)"""
    def make_detector(self, img_last, img_lastlast, fx, fy, cx, cy, depth_scale):
        import oracle_lib as O

        class D:
            def __init__(s):
                s.o = O.DynaDetect(img_last, img_lastlast, fx, fy, cx, cy, depth_scale)

            def DetectDynaArea(s, bgr, depth, ni):
                return s.o.detect(bgr, depth)

            def dilate15(s, dyna):
                return O.dilate15(dyna)
        return D()

    def make_extractor(self, nfeatures, scale_factor, nlevels, ini_th, min_th):
        import oracle_lib as O
        e = O.ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th)
        return lambda gray, mask: e.extract(gray, mask)


@pytest.mark.timeout(600)
def test_config1_twenty_frames_cpu_plumbing(tmp_path):
    """BASELINE.json configs[0] / SURVEY 8d-1: the first 20 frames of the synthetic TUM-shaped stream as a TUM folder (PNG files, association
    list, TUM3.yaml) through the rgbd_tum_noros-shaped loop with the CPU restatement -- parser, PNG codec, settings, frame 0 passed through,
    DetectDynaArea from frame 1 on, dilation, RGB-order gray, ORB, mask PNGs out -- and the same numbers when the detector is driven directly."""
    import oracle_lib as O
    from sindslam_amd.synth import SyntheticStream, TUM3
    n = 20
    bgr, depth = SyntheticStream(seed=12345).frames(0, n)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for i in range(n):
        Hn.write_png(str(tmp_path / "rgb" / f"{i:04d}.png"), bgr[i]); Hn.write_png(str(tmp_path / "depth" / f"{i:04d}.png"), depth[i])
        lines.append(f"{1305031102.0 + i / 30:.6f} rgb/{i:04d}.png {1305031102.0 + i / 30:.6f} depth/{i:04d}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "TUM3.yaml").write_text("%YAML:1.0\nCamera.fx: 535.4\nCamera.fy: 539.2\nCamera.cx: 320.1\nCamera.cy: 247.6\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n"
                                        "ORBextractor.nFeatures: 1500\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 15\nORBextractor.minThFAST: 5\n")
    res = Hn.run_sequence(str(tmp_path / "TUM3.yaml"), str(tmp_path), str(tmp_path / "assoc.txt"), out_dir=str(tmp_path / "out"), verbose=False, backend=_OracleBackend())
    assert len(res) == n and not res[0]["dyna"].any() and len(res[0]["keypoints"]) > 250          # frame 0: all-zero mask, plain ORB
    ref = O.DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"]); orb = O.ORBextractor(1500, 1.2, 8, 15, 5)
    some = False
    for i in range(1, n):
        rd, rl = ref.detect(bgr[i], depth[i])
        assert np.array_equal(res[i]["dyna"], rd) and np.array_equal(res[i]["label"], rl) and np.array_equal(res[i]["mask"], O.dilate15(rd)), i
        rk, rdesc = orb.extract(O.bgr2gray(bgr[i], swap_rb=True), res[i]["mask"])
        assert res[i]["keypoints"].tobytes() == rk.tobytes() and np.array_equal(res[i]["descriptors"], rdesc), i
        assert np.array_equal(Hn.read_png(str(tmp_path / "out" / f"dynaMask_{i:05d}.png")), res[i]["mask"])
        some |= bool((rd == 255).any())
    assert some and set(np.unique(res[5]["dyna"])) <= {0, 125, 255}
