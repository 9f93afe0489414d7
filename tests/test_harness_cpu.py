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
