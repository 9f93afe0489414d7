"""GPU: one long sequence cut into chunks over the streams of the batched pipeline (sindslam_amd/sequence.py) against the sequential
frame loop on the same GPU code: the first chunk is bit-identical, later chunks differ only through the tail state they rebuild in the
warm-up frames -- the mask IoU against the sequential run is reported (seams, median, mean, minimum) and the bulk is bounded."""
import numpy as np
import pytest

from sindslam_amd.sequence import plan_chunks, process_sequence
from sindslam_amd.synth import SyntheticStream, TUM3

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_chunked_sequence_against_sequential_loop():
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    n, S, T, W = 26, 4, 2, 4
    bgr, depth = SyntheticStream(seed=4242).frames(0, n)
    got = process_sequence(bgr, depth, TUM3, streams=S, frames_per_step=T, warmup=W)
    assert got["owned"] == list(range(1, n))
    dd = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    orb = ORBextractor(1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"])
    chunks = plan_chunks(n, S, W); ious = {}
    for f in range(1, n):
        rd, rl = dd.DetectDynaArea(bgr[f], depth[f], f); rm = dd.dilate15(rd)
        b, g, r = bgr[f][..., 0].astype(np.int32), bgr[f][..., 1].astype(np.int32), bgr[f][..., 2].astype(np.int32)
        rk, rdesc = orb(((b * 4899 + g * 9617 + r * 1868 + 8192) >> 14).astype(np.uint8), rm)
        if f < chunks[0].last:                  # chunk 0 IS the sequential run
            assert np.array_equal(got["dyna"][f], rd) and np.array_equal(got["label"][f], rl) and np.array_equal(got["mask"][f], rm), f
            assert got["keypoints"][f].tobytes() == rk.tobytes() and np.array_equal(got["descriptors"][f], rdesc), f
        u = np.logical_or(got["dyna"][f] == 255, rd == 255).sum()
        ious[f] = 1.0 if u == 0 else float(np.logical_and(got["dyna"][f] == 255, rd == 255).sum() / u)
    seams = [c.first for c in chunks[1:]]
    v = np.array([ious[f] for f in range(chunks[1].first, n)])
    print("mask IoU vs the sequential loop at the chunk seams:", {f: round(ious[f], 4) for f in seams}, f" later chunks: median {np.median(v):.4f} mean {v.mean():.4f} min {v.min():.4f}")
    # The tail state (k-means warm labels, PROSAC weights, previous high mask) steers the result: a chunk that rebuilds it in a few warm-up
    # frames returns a valid but not identical mask, and on frames with a small mask the IoU against the sequential run can be low
    # (profiles/tools/seam_iou.py prints the per-frame values for several warm-up lengths).  The bulk of the frames has to agree.
    assert np.median(v) >= 0.97 and v.mean() >= 0.9
    dd.close(); orb.close()
