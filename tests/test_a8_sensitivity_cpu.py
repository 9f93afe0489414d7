"""SURVEY 8 a-8 (cv::findHomography(..., cv::RHO), reference DynaDetect.cc:1235) cannot be pinned: OpenCV's rho.cpp is not restatable bit for
bit offline.  Product and oracle therefore each implement RHO's PUBLISHED scheme (PROSAC + SPRT + LM in float32) and agree bit for bit
(tests/test_host_stages_cpu.py) -- which says nothing about OpenCV's own output.  What CAN be measured is how much the estimator's free
choices matter: from the SAME inter-frame state and the SAME dense flow, the dynamic mask is recomputed with (a) other PRNG seeds of the
scheme (another draw order) and (b) round 1's lighter estimator (PROSAC + least squares + Gauss-Newton, kept in the oracle).  The per-frame
mask IoU against the default is the sensitivity of the path to this stage: the scheme agrees with itself to a few hundredths of a pixel, so
the mask does not hinge on the draw order -- the lighter estimator did (corner shifts of 2-3 px between seeds, IoU down to 0.85), which is
why it was replaced.  The numbers are printed (python -m pytest -s) and bounded."""
import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd.synth import SyntheticStream, TUM3


def iou(a, b):
    u = np.logical_or(a == 255, b == 255).sum()
    return 1.0 if u == 0 else float(np.logical_and(a == 255, b == 255).sum() / u)


def sensitivity(n_frames, seeds, stream_seed=4242):
    bgr, depth = SyntheticStream(seed=stream_seed).frames(0, n_frames)
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    base = O.DynaDetect(bgr[0], bgr[0].copy(), *K)
    out = {"seed": [], "ls": [], "h_seed": [], "h_ls": []}
    for f in range(1, n_frames):
        forks = [("seed", s, base.fork()) for s in seeds] + [("ls", s, base.fork()) for s in (0, seeds[0])]       # all start from the state BEFORE frame f
        dyna, _ = base.detect(bgr[f], depth[f]); dbg = base.debug(); H0 = dbg["H"]
        for kind, s, d in forks:
            d.set_h_estimator(2 if kind == "ls" else 0, s)
            m, _ = d.detect_with_flow(bgr[f], depth[f], dbg["flow_full"]); H = d.debug()["H"]
            out[kind].append(iou(m, dyna))
            # displacement of the image corners between the two homographies (px): how different the models are
            c = np.array([[0, 0, 1], [639, 0, 1], [0, 479, 1], [639, 479, 1]], float).T
            p0 = H0 @ c; p1 = H @ c
            out["h_" + kind].append(float(np.abs(p0[:2] / p0[2] - p1[:2] / p1[2]).max()))
    return {k: np.array(v) for k, v in out.items()}


@pytest.mark.timeout(600)
def test_mask_sensitivity_to_the_homography_estimator():
    r = sensitivity(6, seeds=[11, 12, 13])
    print(f"\\na-8 sensitivity over 5 frames, same state and flow: other seeds of the RHO scheme: mask IoU mean {r['seed'].mean():.4f} min {r['seed'].min():.4f} "
          f"(corner shift median {np.median(r['h_seed']):.3f} max {r['h_seed'].max():.3f} px); round 1's PROSAC + LS estimator: mask IoU mean {r['ls'].mean():.4f} "
          f"min {r['ls'].min():.4f} (corner shift <= {r['h_ls'].max():.3f} px)")
    # the scheme's own draw order must not decide the mask on the bulk of the frames (parity bar of the metric; a frame whose correspondences
    # support two motions about equally well -- here the first one, where n-1 and n-2 are the same image -- stays ambiguous for ANY randomized
    # estimator, OpenCV's included); the lighter estimator's numbers are the documented reason for dropping it
    ns = 3                                   # variants per frame; the first frame's (n-1 == n-2, all-equal sample weights) is the ambiguous one here
    print("per frame, per seed:", np.round(r["seed"].reshape(-1, ns), 4).tolist(), "corner shift px:", np.round(r["h_seed"].reshape(-1, ns), 3).tolist())
    assert r["seed"][ns:].min() >= 0.985 and r["seed"].min() >= 0.85 and np.median(r["h_seed"]) < 0.25


def test_rho_scheme_estimator_recovers_a_known_homography():
    rng = np.random.default_rng(0)
    Ht = np.array([[1.01, 0.002, 3.0], [-0.003, 0.99, -2.0], [1e-6, -2e-6, 1.0]])
    src = rng.uniform(10, 600, (2000, 2)).astype(np.float32)
    p = np.c_[src, np.ones(len(src))] @ Ht.T; dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst += rng.normal(0, 0.3, dst.shape).astype(np.float32); dst[1400:] += rng.uniform(-30, 30, (600, 2)).astype(np.float32)      # 30 % outliers, worst ranked last
    for fn in (O.find_homography, O.find_homography_prosac_ls):
        ok, H = fn(src, dst); assert ok
        q = np.c_[src[:1400], np.ones(1400)] @ H.T; q = q[:, :2] / q[:, 2:]
        assert np.abs(q - p[:1400, :2] / p[:1400, 2:]).max() < 0.5, fn.__name__
