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


def _iou(a, b):
    u = np.logical_or(a == 255, b == 255).sum()
    return 1.0 if u == 0 else float(np.logical_and(a == 255, b == 255).sum() / u)


def _sequential_gpu(bgr, depth):
    """the reference frame loop (rgbd_tum_noros.cc:110-170) on the single-stream GPU classes"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    dd = DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    orb = ORBextractor(1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"]); out = {}
    for f in range(1, len(bgr)):
        rd, rl = dd.DetectDynaArea(bgr[f], depth[f], f); rm = dd.dilate15(rd)
        b, g, r = bgr[f][..., 0].astype(np.int32), bgr[f][..., 1].astype(np.int32), bgr[f][..., 2].astype(np.int32)
        rk, rdesc = orb(((b * 4899 + g * 9617 + r * 1868 + 8192) >> 14).astype(np.uint8), rm)
        out[f] = (rd, rl, rm, rk, rdesc)
    dd.close(); orb.close()
    return out


@pytest.mark.timeout(900)
def test_exact_sequence_equals_sequential_loop():
    """in-order mode (phase A batched, both tail chains strictly in frame order, pipelined steps, a remainder step on a second handle):
    every output of every frame equals the sequential loop bit for bit"""
    from sindslam_amd.sequence import process_sequence_exact
    n = 27
    bgr, depth = SyntheticStream(seed=4242).frames(0, n)
    got = process_sequence_exact(bgr, depth, TUM3, frames_per_step=8)          # steps 8, 8, 8 + remainder 2
    ref = _sequential_gpu(bgr, depth)
    assert got["owned"] == list(range(1, n))
    for f in range(1, n):
        rd, rl, rm, rk, rdesc = ref[f]
        assert np.array_equal(got["dyna"][f], rd) and np.array_equal(got["label"][f], rl) and np.array_equal(got["mask"][f], rm), f
        assert got["keypoints"][f].tobytes() == rk.tobytes() and np.array_equal(got["descriptors"][f], rdesc), f


@pytest.mark.timeout(1500)
def test_exact_and_chunked_modes_against_the_oracle():
    """>= 60 frames against the ORACLE's sequential run (reference state roll DynaDetect.cc:1660-1664): the in-order mode has to meet
    the IoU >= 0.99 bar on every frame; the chunked (throughput) mode is a different, documented trade -- its numbers are printed."""
    import oracle_lib as O
    from sindslam_amd.sequence import process_sequence_exact
    n = 62
    bgr, depth = SyntheticStream(seed=777).frames(0, n)
    ora = O.DynaDetect(bgr[0], bgr[0].copy(), TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    ref = {f: ora.detect(bgr[f], depth[f])[0] for f in range(1, n)}
    ex = process_sequence_exact(bgr, depth, TUM3, frames_per_step=16, want_keypoints=False)
    e = np.array([_iou(ex["dyna"][f], ref[f]) for f in range(1, n)])
    ch = process_sequence(bgr, depth, TUM3, streams=4, frames_per_step=4, warmup=5, want_keypoints=False)
    c = np.array([_iou(ch["dyna"][f], ref[f]) for f in range(1, n)])
    ch2 = process_sequence(bgr, depth, TUM3, streams=2, frames_per_step=4, warmup=20, want_keypoints=False)
    c2 = np.array([_iou(ch2["dyna"][f], ref[f]) for f in range(1, n)])
    print(f"mask IoU vs the oracle's sequential run over {n - 1} frames: exact mode mean {e.mean():.4f} min {e.min():.4f}; "
          f"chunked mode (4 chunks, warm-up 5) mean {c.mean():.4f} median {np.median(c):.4f} min {c.min():.4f}; "
          f"chunked mode (2 chunks, warm-up 20) mean {c2.mean():.4f} min {c2.min():.4f}, frames below 0.99: {(c2 < 0.99).sum()}")
    assert e.min() >= 0.99, e
    assert np.median(c) >= 0.97
    assert c2.mean() >= 0.98 and np.median(c2) >= 0.99          # a long enough warm-up re-synchronises the chunk state


@pytest.mark.timeout(900)
def test_exact_sequence_two_ranks_hand_the_state_over(tmp_path):
    """two processes on this one card (gloo for the state send / recv): rank 0 owns the first half of the frames, rank 1 the second and
    continues from rank 0's state blob -- together they reproduce the one-rank in-order run bit for bit"""
    import os, subprocess, sys
    from sindslam_amd.sequence import process_sequence_exact
    n = 21
    one = process_sequence_exact(*SyntheticStream(seed=99).frames(0, n), TUM3, frames_per_step=6, want_keypoints=False)
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "seq_exact_worker.py"), str(n), str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        for f in z["owned"]:
            assert np.array_equal(z["dyna"][f], one["dyna"][f]) and np.array_equal(z["label"][f], one["label"][f]), (r, int(f))
    assert sorted(np.concatenate([np.load(tmp_path / f"rank{r}.npz")["owned"] for r in range(2)]).tolist()) == list(range(1, n))
