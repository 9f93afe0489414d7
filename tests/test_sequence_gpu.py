"""GPU: one long sequence cut into lock-step chunks over the streams of the batched pipeline (sindslam_amd/sequence.py VerifiedChunks: speculate, verify the
chunk seams by state fingerprints, repair the chunks whose rebuilt state is not the sequential one) against the sequential frame loop -- on the same GPU
code and on the ORACLE: every output of every frame is equal.  Plus the in-order mode (process_sequence_exact)."""
import numpy as np
import pytest

from sindslam_amd.sequence import process_sequence
from sindslam_amd.synth import SyntheticStream, TUM3

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_chunked_sequence_against_sequential_loop():
    """4 chunks, warm-up 4 (too short to re-synchronise every seam by itself): after verification / repair every frame equals the sequential loop"""
    n, S, T, W = 26, 4, 2, 4
    bgr, depth = SyntheticStream(seed=4242).frames(0, n)
    st = {}
    got = process_sequence(bgr, depth, TUM3, streams=S, frames_per_step=T, warmup=W, repair_streams=3, repair_frames_per_step=2, stats=st)
    assert got["owned"] == list(range(1, n))
    ref = _sequential_gpu(bgr, depth)
    print("chunk seams:", {k: v for k, v in st.items() if k != "plan"})
    for f in range(1, n):
        rd, rl, rm, rk, rdesc = ref[f]
        assert np.array_equal(got["dyna"][f], rd) and np.array_equal(got["label"][f], rl) and np.array_equal(got["mask"][f], rm), f
        assert got["keypoints"][f].tobytes() == rk.tobytes() and np.array_equal(got["descriptors"][f], rdesc), f
    assert st["seams"] == 3


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
    """>= 60 frames against the ORACLE's sequential run (reference state roll DynaDetect.cc:1660-1664): the in-order mode and the chunked mode (4 chunks with
    a 5-frame warm-up, 2 chunks with a 20-frame warm-up) return the oracle's imgDyna on every frame"""
    import oracle_lib as O
    from sindslam_amd.sequence import process_sequence_exact
    n = 62
    bgr, depth = SyntheticStream(seed=777).frames(0, n)
    ref = O.sequence_run(bgr, depth, TUM3, threads=8, want_orb=False)
    ex = process_sequence_exact(bgr, depth, TUM3, frames_per_step=16, want_keypoints=False)
    s1, s2 = {}, {}
    ch = process_sequence(bgr, depth, TUM3, streams=4, frames_per_step=4, warmup=5, want_keypoints=False, stats=s1)
    ch2 = process_sequence(bgr, depth, TUM3, streams=2, frames_per_step=4, warmup=20, want_keypoints=False, stats=s2)
    print("4 chunks / warm-up 5:", {k: v for k, v in s1.items() if k != "plan"}); print("2 chunks / warm-up 20:", {k: v for k, v in s2.items() if k != "plan"})
    for f in range(1, n):
        for name, got in (("in-order", ex), ("4 chunks", ch), ("2 chunks", ch2)):
            assert np.array_equal(got["dyna"][f], ref["dyna"][f]) and np.array_equal(got["label"][f], ref["label"][f]), (name, f)


@pytest.mark.timeout(2400)
def test_verified_chunks_equal_the_oracle_on_125_frames():
    """125 frames on 5 chunks (warm-up 6: short on purpose, so that seams mismatch and runners repair them) against the ORACLE's sequential loop: imgDyna,
    imgLabel, the dilated mask, ORB keypoints and descriptors of every frame are equal"""
    import oracle_lib as O
    n = 126
    bgr, depth = SyntheticStream(seed=2024).frames(0, n)
    st = {}
    got = process_sequence(bgr, depth, TUM3, streams=5, frames_per_step=4, warmup=6, repair_streams=4, repair_frames_per_step=4, stats=st, retain_frames=6)      # replay (tails only) on the first frames, then the repair pipeline
    print("5 chunks / warm-up 6:", {k: v for k, v in st.items() if k != "plan"}, st["plan"].chunks)
    assert got["owned"] == list(range(1, n)) and st["seams"] == 4
    ref = O.sequence_run(bgr, depth, TUM3, threads=12)
    for f in range(1, n):
        assert np.array_equal(got["dyna"][f], ref["dyna"][f]) and np.array_equal(got["label"][f], ref["label"][f]) and np.array_equal(got["mask"][f], ref["mask"][f]), f
        assert got["keypoints"][f].tobytes() == ref["keypoints"][f].tobytes() and np.array_equal(got["descriptors"][f], ref["descriptors"][f]), f


@pytest.mark.timeout(900)
def test_verified_chunks_forced_mismatch_no_warmup():
    """warm-up 0: every chunk after the first starts from an empty state, every seam is a mismatch and is repaired; without verification the same run differs"""
    n = 31
    bgr, depth = SyntheticStream(seed=4242).frames(0, n)
    st = {}
    got = process_sequence(bgr, depth, TUM3, streams=3, frames_per_step=2, warmup=0, repair_streams=2, repair_frames_per_step=3, stats=st, want_keypoints=False, retain_frames=0)
    raw = process_sequence(bgr, depth, TUM3, streams=3, frames_per_step=2, warmup=0, verify=False, want_keypoints=False)
    ref = _sequential_gpu(bgr, depth)
    print("3 chunks / no warm-up:", {k: v for k, v in st.items() if k != "plan"})
    assert st["mismatched_seams"] == 2 and st["rounds"] >= 1 and st["repair_frames"] > 0
    differ = 0
    for f in range(1, n):
        assert np.array_equal(got["dyna"][f], ref[f][0]) and np.array_equal(got["label"][f], ref[f][1]) and np.array_equal(got["mask"][f], ref[f][2]), f
        differ += int(not np.array_equal(raw["dyna"][f], ref[f][0]) or not np.array_equal(raw["label"][f], ref[f][1]))
    assert differ > 0, "the unverified run was expected to differ behind a cold seam"


@pytest.mark.timeout(900)
def test_verified_chunks_two_ranks_forced_mismatch(tmp_path):
    """two processes on this one card (gloo): 2 x 2 chunks with a 1-frame warm-up -- the seam between the ranks mismatches, rank 1 receives rank 0's end-state
    blob and repairs its first chunk; together the ranks reproduce the sequential loop bit for bit"""
    import json, os, subprocess, sys
    n = 33
    bgr, depth = SyntheticStream(seed=99).frames(0, n)
    ref = _sequential_gpu(bgr, depth)
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "seq_verified_worker.py"), str(n), str(tmp_path), "1"], env=env))
    for p in procs:
        assert p.wait(timeout=800) == 0
    owned = []; mism = 0
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz"); st = json.loads(str(z["stats"])); print(f"rank {r}:", st); mism = max(mism, st["mismatched_seams"])
        for f in z["owned"]:
            assert np.array_equal(z["dyna"][f], ref[int(f)][0]) and np.array_equal(z["label"][f], ref[int(f)][1]) and np.array_equal(z["mask"][f], ref[int(f)][2]), (r, int(f))
        owned += z["owned"].tolist()
    assert sorted(owned) == list(range(1, n)) and mism >= 1


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


@pytest.mark.timeout(900)
def test_verified_chunks_1280x720_three_level_pyramid_and_bonn():
    """the other BASELINE configs through the chunked mode: (i) 1280 x 720, D455 intrinsics x 2, depth factor 1000, FAST 20 / 7, 3-level flow pyramid, BGR2GRAY for ORB;
    (ii) 640 x 480 with the Bonn intrinsics and FAST 20 / 7 -- 3 chunks with a 2-frame warm-up each, against the in-order mode on the same pipeline configuration"""
    from sindslam_amd.pipeline import Pipeline
    from sindslam_amd.sequence import process_sequence_exact
    from sindslam_amd.synth import BONN, D455
    n = 20
    s = SyntheticStream(width=1280, height=720, intr=D455, motion_scale=0.5)
    bgr, depth = s.frames(0, n)
    intr = dict(D455, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy)
    mk = lambda S_, T_: Pipeline(S_, T_, 1280, 720, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], 1500, 1.2, 8, intr["ini_th"], intr["min_th"],
                                 orb_gray_rgb_order=0, flow_max_levels=3)
    st = {}
    got = process_sequence(bgr, depth, intr, streams=3, frames_per_step=3, warmup=2, repair_streams=2, repair_frames_per_step=2, stats=st, pipeline_factory=mk, retain_frames=4)
    print("1280x720:", {k: v for k, v in st.items() if k != "plan"})
    # reference: one stream, one frame per step, synchronous -- the sequential loop on the same pipeline configuration
    one = mk(1, 1); one.prime(0, bgr[0], bgr[0])
    for f in range(1, n):
        one.process(bgr[f][None, None], depth[f][None, None])
        assert np.array_equal(got["dyna"][f], one.dyna[0, 0]) and np.array_equal(got["label"][f], one.label[0, 0]) and np.array_equal(got["mask"][f], one.mask[0, 0]), f
        k, d = one.keypoints(0, 0)
        assert got["keypoints"][f].tobytes() == k.tobytes() and np.array_equal(got["descriptors"][f], d), f
    one.close()
    assert st["seams"] == 2
    bgr, depth = SyntheticStream(seed=5, intr=BONN).frames(0, n)
    st = {}
    got = process_sequence(bgr, depth, BONN, streams=3, frames_per_step=3, warmup=2, repair_streams=2, repair_frames_per_step=2, stats=st, want_keypoints=False)
    ex = process_sequence_exact(bgr, depth, BONN, frames_per_step=8, want_keypoints=False)
    print("bonn:", {k: v for k, v in st.items() if k != "plan"})
    for f in range(1, n):
        assert np.array_equal(got["dyna"][f], ex["dyna"][f]) and np.array_equal(got["label"][f], ex["label"][f]) and np.array_equal(got["mask"][f], ex["mask"][f]), f


# ---- the same through the C++ driver behind sind_seq_* (no Python in the loop, no torch.distributed between the ranks)
@pytest.mark.timeout(2400)
def test_cabi_verified_chunks_equal_the_oracle_on_125_frames():
    """125 frames on 5 chunks (warm-up 6: short on purpose, so that seams mismatch and runners repair them) through sind_seq_run against the ORACLE's sequential loop:
    imgDyna, imgLabel, the dilated mask, ORB keypoints and descriptors of every frame are equal"""
    import oracle_lib as O
    from sindslam_amd.seq import run_sequence
    n = 126
    bgr, depth = SyntheticStream(seed=2024).frames(0, n)
    st = {}
    got = run_sequence(bgr, depth, TUM3, streams=5, frames_per_step=4, warmup=6, repair_streams=4, repair_frames_per_step=4, stats=st, retain_frames=6)
    print("C ABI, 5 chunks / warm-up 6:", st)
    assert got["owned"] == list(range(1, n)) and st["seams"] == 4 and st["mismatched_seams"] >= 1
    ref = O.sequence_run(bgr, depth, TUM3, threads=12)
    for f in range(1, n):
        assert np.array_equal(got["dyna"][f], ref["dyna"][f]) and np.array_equal(got["label"][f], ref["label"][f]) and np.array_equal(got["mask"][f], ref["mask"][f]), f
        assert got["keypoints"][f].tobytes() == ref["keypoints"][f].tobytes() and np.array_equal(got["descriptors"][f], ref["descriptors"][f]), f


@pytest.mark.timeout(900)
def test_cabi_forced_mismatch_no_warmup_equals_the_python_driver():
    """warm-up 0 through sind_seq_run: every seam mismatches and is repaired; results and statistics equal the Python driver's on the same input"""
    from sindslam_amd.seq import run_sequence
    n = 31
    bgr, depth = SyntheticStream(seed=4242).frames(0, n)
    st = {}; pst = {}
    got = run_sequence(bgr, depth, TUM3, streams=3, frames_per_step=2, warmup=0, repair_streams=2, repair_frames_per_step=3, stats=st, want_keypoints=False, retain_frames=0)
    py = process_sequence(bgr, depth, TUM3, streams=3, frames_per_step=2, warmup=0, repair_streams=2, repair_frames_per_step=3, stats=pst, want_keypoints=False, retain_frames=0)
    assert st["mismatched_seams"] == 2 and st["rounds"] >= 1 and st["repair_frames"] > 0
    for k in ("seams", "mismatched_seams", "rounds", "runners", "repaired_chunks", "repair_frames", "overridden_frames", "max_frames_to_converge"):
        assert st[k] == pst[k], (k, st[k], pst[k])
    for f in range(1, n):
        assert np.array_equal(got["dyna"][f], py["dyna"][f]) and np.array_equal(got["label"][f], py["label"][f]) and np.array_equal(got["mask"][f], py["mask"][f]), f


@pytest.mark.parametrize("world,warm", [(2, 1), (3, 0)])
@pytest.mark.timeout(1200)
def test_cabi_ranks_forced_mismatch_over_tcp(tmp_path, world, warm):
    """two / three PROCESSES on this one card, 2 chunks each, the exchange over loopback TCP (sind_seq_net_tcp): the seams between the ranks mismatch, a rank receives its
    predecessor's end-state blob (the middle rank of three also sends its own) and repairs; together the ranks reproduce the sequential loop bit for bit"""
    import json, os, subprocess, sys
    n = 33
    bgr, depth = SyntheticStream(seed=99).frames(0, n)
    ref = _sequential_gpu(bgr, depth)
    port = 33000 + (os.getpid() * 3 + world) % 20000
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "seq_cabi_worker.py"), str(n), str(tmp_path), str(warm), str(r), str(world), str(port)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=1000) == 0
    owned = []; mism = 0
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz"); st = json.loads(str(z["stats"])); print(f"rank {r}:", st); mism = max(mism, st["mismatched_seams"])
        for f in z["owned"]:
            assert np.array_equal(z["dyna"][f], ref[int(f)][0]) and np.array_equal(z["label"][f], ref[int(f)][1]) and np.array_equal(z["mask"][f], ref[int(f)][2]), (r, int(f))
        owned += z["owned"].tolist()
    assert sorted(owned) == list(range(1, n)) and mism >= world - 1
