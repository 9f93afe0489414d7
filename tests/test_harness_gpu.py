"""GPU: the rgbd_tum_noros-shaped harness on a TUM-style folder equals the direct class API."""
import os

import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd import harness as Hn
from sindslam_amd.synth import TUM3

pytestmark = pytest.mark.gpu


def test_sequence_folder(tmp_path, frames):
    bgr, depth = frames
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for i in range(4):
        Hn.write_png(str(tmp_path / "rgb" / f"{i}.png"), bgr[i]); Hn.write_png(str(tmp_path / "depth" / f"{i}.png"), depth[i])
        lines.append(f"{1000.0 + i / 30:.6f} rgb/{i}.png {1000.0 + i / 30:.6f} depth/{i}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "TUM3.yaml").write_text("%YAML:1.0\nCamera.fx: 535.4\nCamera.fy: 539.2\nCamera.cx: 320.1\nCamera.cy: 247.6\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n"
                                        "ORBextractor.nFeatures: 1500\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 15\nORBextractor.minThFAST: 5\n")
    res = Hn.run_sequence(str(tmp_path / "TUM3.yaml"), str(tmp_path), str(tmp_path / "assoc.txt"), out_dir=str(tmp_path / "out"), verbose=False)
    assert len(res) == 4 and not res[0]["mask"].any()                          # frame 0: all-zero mask (rgbd_tum_noros.cc:100-116)
    ref = O.DynaDetect(bgr[0], bgr[0], TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], 5000.0); orb = O.ORBextractor(1500, 1.2, 8, 15, 5)
    for i in range(1, 4):
        rd, rl = ref.detect(bgr[i], depth[i])
        u = np.logical_or(res[i]["dyna"] == 255, rd == 255).sum()
        assert u == 0 or np.logical_and(res[i]["dyna"] == 255, rd == 255).sum() / u >= 0.99
        rk, rdesc = orb.extract(O.bgr2gray(bgr[i], swap_rb=True), res[i]["mask"])
        assert res[i]["keypoints"].tobytes() == rk.tobytes() and np.array_equal(res[i]["descriptors"], rdesc)
        assert np.array_equal(Hn.read_png(str(tmp_path / "out" / f"dynaMask_{i:05d}.png")), res[i]["mask"])


@pytest.mark.timeout(900)
def test_sequence_folder_chunked_equals_the_frame_loop(tmp_path):
    """a 21-frame TUM-style folder through the harness twice: the reference-shaped frame loop (run_sequence) and the offline mode on 3 verified chunks
    (run_sequence_chunked) -- every output of every frame is equal, frame 0 (passed through with an empty mask) included, and so are the PNG files written"""
    from sindslam_amd.synth import SyntheticStream
    n = 21
    bgr, depth = SyntheticStream(seed=321).frames(0, n)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for i in range(n):
        Hn.write_png(str(tmp_path / "rgb" / f"{i}.png"), bgr[i]); Hn.write_png(str(tmp_path / "depth" / f"{i}.png"), depth[i])
        lines.append(f"{1000.0 + i / 30:.6f} rgb/{i}.png {1000.0 + i / 30:.6f} depth/{i}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "TUM3.yaml").write_text("%YAML:1.0\nCamera.fx: 535.4\nCamera.fy: 539.2\nCamera.cx: 320.1\nCamera.cy: 247.6\nCamera.RGB: 1\nDepthMapFactor: 5000.0\n"
                                        "ORBextractor.nFeatures: 1500\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 15\nORBextractor.minThFAST: 5\n")
    a = Hn.run_sequence(str(tmp_path / "TUM3.yaml"), str(tmp_path), str(tmp_path / "assoc.txt"), out_dir=str(tmp_path / "out_a"), verbose=False)
    b = Hn.run_sequence_chunked(str(tmp_path / "TUM3.yaml"), str(tmp_path), str(tmp_path / "assoc.txt"), out_dir=str(tmp_path / "out_b"), verbose=True, chunks=3, frames_per_step=3, warmup=2)
    assert len(a) == len(b) == n
    for i in range(n):
        for k in ("dyna", "label", "mask", "descriptors"):
            assert np.array_equal(a[i][k], b[i][k]), (i, k)
        assert a[i]["keypoints"].tobytes() == b[i]["keypoints"].tobytes() and a[i]["timestamp"] == b[i]["timestamp"], i
        assert np.array_equal(Hn.read_png(str(tmp_path / "out_a" / f"dynaMask_{i:05d}.png")), Hn.read_png(str(tmp_path / "out_b" / f"dynaMask_{i:05d}.png")))
