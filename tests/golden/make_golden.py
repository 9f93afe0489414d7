"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so) on the seeded synthetic stream.

The reference repository ships no golden vectors and cannot be built or imported here (OpenCV 4.2.0 + contrib, PCL,
Eigen are absent), so these fixtures pin the ORACLE, not the reference: they guard the restatement against accidental
change and give the GPU tests a second, file-based comparison point.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from sindslam_amd.synth import SyntheticStream, TUM3  # noqa: E402


# fx fy cx cy k1 k2 p1 p2 k3 bf depthMapFactor (the reference's Examples/RGB-D/TUM3.yaml and TUM1.yaml values)
FRAME_CALIBS = {
    "tum3": [535.4, 539.2, 320.1, 247.6, 0.0, 0.0, 0.0, 0.0, 0.0, 40.0, 1.0 / 5000.0],
    "tum1": [517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0, 1.0 / 5000.0],
}


def main():
    s = SyntheticStream(seed=12345)
    bgr, depth = s.frames(0, 4)
    g = [O.bgr2gray(b) for b in bgr]
    # 1. flow stage on a small pair (96x72): DeepFlow, then refinement of the negated flow
    a, b = O.resize_u8(g[2], 96, 72), O.resize_u8(g[0], 96, 72)
    f = O.deepflow(a, b)
    ru, rv = O.varref(a.astype(np.float32), b.astype(np.float32), -f[..., 0], -f[..., 1])
    np.savez_compressed(os.path.join(HERE, "flow_96x72.npz"), i0=a, i1=b, deep=f, refined=np.stack([ru, rv]))
    # 2. ORB on one 640x480 frame (TUM3 settings) with and without a dynamic mask
    orb = O.ORBextractor(1500, 1.2, 8, 15, 5)
    k0, d0 = orb.extract(g[2])
    mask = np.zeros((480, 640), np.uint8); mask[:, :320] = 255
    k1, d1 = orb.extract(g[2], mask)
    np.savez_compressed(os.path.join(HERE, "orb_frame2.npz"), kps=k0, desc=d0, kps_masked=k1, desc_masked=d1)
    # 3. DynaDetect on frames 2, 3 (primed with 1, 0): thresholds, homography and bit-packed output masks
    dd = O.DynaDetect(bgr[1], bgr[0], TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    out = {}
    for t in (2, 3):
        dy, lb = dd.detect(bgr[t], depth[t]); dbg = dd.debug()
        out[f"dyna255_{t}"] = np.packbits(dy == 255); out[f"dyna125_{t}"] = np.packbits(dy == 125); out[f"label_{t}"] = lb
        out[f"H_{t}"] = dbg["H"]; out[f"thr_{t}"] = dbg["thr"]; out[f"hist_{t}"] = dbg["hist"]; out[f"info_{t}"] = dbg["info"]
        out[f"occ1_{t}"] = np.packbits(dbg["occ1"] > 0); out[f"occ2_{t}"] = np.packbits(dbg["occ2"] > 0)
    np.savez_compressed(os.path.join(HERE, "dyna_frames23.npz"), **out)
    # 4. primitive vectors
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), gray2=g[2][::8, ::8].copy(), gray_min=O.resize_u8(g[2], 384, 288)[::6, ::6].copy(),
                        rng=O.rng_gaussian(12345, 0.5, 64), blur=O.gaussian_blur_u8(g[2][:64, :64].copy()), pyr_sizes=np.array(O.deepflow_levels(384, 288)))
    # 5. Frame post-ORB steps on the keypoints of (2): TUM3 (no distortion) and TUM1 (radial + tangential distortion) calibrations
    frame_fixture(k0, depth[2])
    print("golden fixtures written to", HERE)


def frame_fixture(kps, depth2):
    fp = {}
    for name, cal in FRAME_CALIBS.items():
        r = O.frame_post_orb(cal, kps["x"], kps["y"], depth2)
        for key, val in r.items():
            fp[f"{name}_{key}"] = val
    np.savez_compressed(os.path.join(HERE, "frame_post.npz"), **fp)


if __name__ == "__main__":
    if sys.argv[1:] == ["frame"]:       # only (5), from the committed ORB fixture
        frame_fixture(np.load(os.path.join(HERE, "orb_frame2.npz"))["kps"], SyntheticStream(seed=12345).frames(2, 1)[1][0])
    else:
        main()
