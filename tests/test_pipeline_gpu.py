"""GPU parity: the batched multi-stream pipeline equals independent single-stream oracle runs."""
import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd.synth import TUM3

pytestmark = pytest.mark.gpu


def test_pipeline_two_streams(frames):
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames                       # 6 frames
    S, T = 2, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    # stream 0 = the sequence, stream 1 = its horizontal mirror
    sb = np.stack([bgr, bgr[:, :, ::-1]]); sd = np.stack([depth, depth[:, :, ::-1]])
    pipe = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1)
    refs = [O.DynaDetect(np.ascontiguousarray(sb[s, 1]), np.ascontiguousarray(sb[s, 0]), *K) for s in range(S)]
    orb_ref = O.ORBextractor(1500, 1.2, 8, 15, 5)
    for s in range(S):
        pipe.prime(s, sb[s, 1], sb[s, 0])
    for step in range(2):                     # two steps exercise the history roll
        lo = 2 + step * T
        pipe.process(sb[:, lo:lo + T], sd[:, lo:lo + T])
        for s in range(S):
            for t in range(T):
                rd, rl = refs[s].detect(np.ascontiguousarray(sb[s, lo + t]), np.ascontiguousarray(sd[s, lo + t]))
                gd = pipe.dyna[s, t]
                u = np.logical_or(gd == 255, rd == 255).sum()
                iou = 1.0 if u == 0 else np.logical_and(gd == 255, rd == 255).sum() / u
                assert iou >= 0.99, (step, s, t, iou)
                assert np.array_equal(pipe.mask[s, t], O.dilate15(gd))
                gray = O.bgr2gray(np.ascontiguousarray(sb[s, lo + t]), swap_rb=True)      # Camera.RGB: 1 quirk
                rk, rdesc = orb_ref.extract(gray, pipe.mask[s, t])
                k, d = pipe.keypoints(s, t)
                assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc), (step, s, t)
    st = pipe.stats()
    assert st["sor_launches"] + st["sor_other_launches"] > 0 and st["sor_ms"] + st["sor_other_ms"] > 0       # (a two-stream step has no streaming launches: those count under sor_*)
    pipe.close()


def test_pipelined_submit_equals_sync(frames):
    """submit/flush (phase A of step i overlapping the tails of step i-1) returns exactly the synchronous results, one call later"""
    import torch
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    S, T = 2, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    sb = np.stack([bgr, bgr[:, ::-1]]); sd = np.stack([depth, depth[:, ::-1]])
    ref = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5); pip = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
    for s in range(S):
        ref.prime(s, sb[s, 1], sb[s, 0]); pip.prime(s, sb[s, 1], sb[s, 0])
    expect = []
    for step in range(2):
        lo = 2 + step * T
        ref.process(sb[:, lo:lo + T], sd[:, lo:lo + T])
        expect.append((ref.dyna.copy(), ref.label.copy(), ref.mask.copy(), ref.nkp.copy(), ref.kps.copy(), ref.desc.copy()))
    dev = [(torch.from_numpy(np.ascontiguousarray(sb[:, 2 + i * T: 4 + i * T])).cuda(), torch.from_numpy(np.ascontiguousarray(sd[:, 2 + i * T: 4 + i * T]).view(np.int16)).cuda()) for i in range(2)]
    got = []
    assert pip.submit_dev(dev[0][0].data_ptr(), dev[0][1].data_ptr()) is False
    assert pip.submit_dev(dev[1][0].data_ptr(), dev[1][1].data_ptr()) is True
    got.append((pip.dyna.copy(), pip.label.copy(), pip.mask.copy(), pip.nkp.copy(), pip.kps.copy(), pip.desc.copy()))
    assert pip.flush() is True
    got.append((pip.dyna.copy(), pip.label.copy(), pip.mask.copy(), pip.nkp.copy(), pip.kps.copy(), pip.desc.copy()))
    assert pip.flush() is False
    for e, g in zip(expect, got):
        for a, b in zip(e[:4], g[:4]):
            assert np.array_equal(a, b)
        for k in range(S * T):
            n = e[3][k]; assert e[4][k, :n].tobytes() == g[4][k, :n].tobytes() and np.array_equal(e[5][k, :n], g[5][k, :n])
    ref.close(); pip.close()


def test_depth_ahead_schedule_equals_default(frames):
    """the depth half of the tails run underneath the dense flow (sind_pipe_set_depth_ahead) returns bit-identical results"""
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    S, T = 3, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    sb = np.stack([bgr, bgr[:, ::-1], bgr[:, :, ::-1]]); sd = np.stack([depth, depth[:, ::-1], depth[:, :, ::-1]])
    ref = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5); pip = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
    pip.set_depth_ahead(True)
    for s in range(S):
        ref.prime(s, sb[s, 1], sb[s, 0]); pip.prime(s, sb[s, 1], sb[s, 0])
    for step in range(2):                     # the second step starts from rolled state (k-means warm labels, previous masks)
        lo = 2 + step * T
        ref.process(sb[:, lo:lo + T], sd[:, lo:lo + T]); pip.process(sb[:, lo:lo + T], sd[:, lo:lo + T])
        assert np.array_equal(ref.dyna, pip.dyna) and np.array_equal(ref.label, pip.label) and np.array_equal(ref.mask, pip.mask)
        assert np.array_equal(ref.nkp, pip.nkp)
        for k in range(S * T):
            n = ref.nkp[k]; assert ref.kps[k, :n].tobytes() == pip.kps[k, :n].tobytes() and np.array_equal(ref.desc[k, :n], pip.desc[k, :n])
        assert (ref.label > 0).any()
    ref.close(); pip.close()


def test_single_frame_steps_and_the_state_blob(frames):
    """S = 1, T = 1 (the smallest batch: the priming scratch must hold two frames) against the single-stream DynaDetect class, and a second
    handle that continues the stream from the first one's state blob (sind_pipe_get_state / set_state) instead of its own history"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    dd = DynaDetect(bgr[1], bgr[0], *K)
    a = Pipeline(1, 1, 640, 480, *K, 1500, 1.2, 8, 15, 5); a.prime(0, bgr[1], bgr[0])
    for f in (2, 3):
        rd, rl = dd.DetectDynaArea(bgr[f], depth[f], f)
        a.process(bgr[None, None, f], depth[None, None, f])
        assert np.array_equal(a.dyna[0, 0], rd) and np.array_equal(a.label[0, 0], rl), f
    blob = a.get_state(0)
    assert blob.size == Pipeline.state_bytes_for(640, 480) == a.get_state_bytes()
    b = Pipeline(1, 2, 640, 480, *K, 1500, 1.2, 8, 15, 5); b.prime(0, bgr[3], bgr[2]); b.set_state(0, blob)
    b.process(bgr[None, 4:6], depth[None, 4:6])
    for t, f in enumerate((4, 5)):
        rd, rl = dd.DetectDynaArea(bgr[f], depth[f], f)
        assert np.array_equal(b.dyna[0, t], rd) and np.array_equal(b.label[0, t], rl), f
    # without the blob the same frames give another (valid) answer: the state matters
    c = Pipeline(1, 2, 640, 480, *K, 1500, 1.2, 8, 15, 5); c.prime(0, bgr[3], bgr[2]); c.process(bgr[None, 4:6], depth[None, 4:6])
    assert not np.array_equal(c.label, b.label)
    dd.close(); a.close(); b.close(); c.close()


def test_pipelined_depth_ahead_equals_sync(frames):
    """the schedule of the in-order sequence mode: submit / flush with depth-ahead (depth chain of step i+1 next to the flow chain of step i,
    on separate tail objects) returns the synchronous results"""
    import torch
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    S, T = 1, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    ref = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5); pip = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
    ref.prime(0, bgr[1], bgr[0]); pip.prime(0, bgr[1], bgr[0]); pip.set_depth_ahead(True)
    expect = []
    for step in range(2):
        lo = 2 + step * T
        ref.process(bgr[None, lo:lo + T], depth[None, lo:lo + T]); expect.append((ref.dyna.copy(), ref.label.copy(), ref.mask.copy()))
    dev = [(torch.from_numpy(np.ascontiguousarray(bgr[None, 2 + i * T: 4 + i * T])).cuda(), torch.from_numpy(np.ascontiguousarray(depth[None, 2 + i * T: 4 + i * T]).view(np.int16)).cuda()) for i in range(2)]
    got = []
    assert pip.submit_dev(dev[0][0].data_ptr(), dev[0][1].data_ptr()) is False
    assert pip.submit_dev(dev[1][0].data_ptr(), dev[1][1].data_ptr()) is True
    got.append((pip.dyna.copy(), pip.label.copy(), pip.mask.copy()))
    assert pip.flush() is True
    got.append((pip.dyna.copy(), pip.label.copy(), pip.mask.copy()))
    for e, g in zip(expect, got):
        for x, y in zip(e, g):
            assert np.array_equal(x, y)
    ref.close(); pip.close()


def test_batched_caloccluded_stage_equals_the_per_frame_chain(frames, monkeypatch):
    """the GPU half of CalOccluded run once per chunk of frames at the start of a step (default for >= 4 frames per step; here in chunks of 3, so that
    a ragged last chunk is covered) returns what the per-frame launches of the worker tasks return (SIND_OCC_BATCH=0), output for output"""
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    S, T = 2, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    sb = np.stack([bgr, bgr[:, :, ::-1]]); sd = np.stack([depth, depth[:, :, ::-1]])
    out = {}
    for batch in ("1", "0"):
        monkeypatch.setenv("SIND_OCC_BATCH", batch); monkeypatch.setenv("SIND_OCC_CHUNK", "3")
        pipe = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
        for s in range(S):
            pipe.prime(s, sb[s, 1], sb[s, 0])
        res = []
        for step in range(2):
            lo = 2 + step * T
            pipe.process(sb[:, lo:lo + T], sd[:, lo:lo + T])
            res.append((pipe.dyna.copy(), pipe.label.copy(), pipe.mask.copy(), pipe.nkp.copy()))
        out[batch] = res; pipe.close()
    for a, b in zip(out["1"], out["0"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert (out["1"][1][0] == 255).any()


def test_region_grow_share_does_not_change_the_results(frames):
    """the PEAC region grow of CalOccluded on the GPU (k_peac_grow), on the host, or split frame by frame: same masks, labels and keypoints"""
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    S, T = 2, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    sb = np.stack([bgr, bgr[:, :, ::-1]]); sd = np.stack([depth, depth[:, ::-1]])
    outs = []
    for q in (0, 4, 1, -1):
        pipe = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
        pipe.set_grow_share(q)
        assert pipe.grow_share() == (q if q >= 0 else pipe.grow_share())
        for s in range(S):
            pipe.prime(s, sb[s, 1], sb[s, 0])
        got = []
        for step in range(2):
            lo = 2 + step * T
            pipe.process(sb[:, lo:lo + T], sd[:, lo:lo + T])
            got.append((pipe.dyna.copy(), pipe.label.copy(), pipe.mask.copy(), pipe.nkp.copy(), pipe.kps.copy(), pipe.desc.copy()))
        assert 0 <= pipe.grow_share() <= 4
        pipe.close(); outs.append(got)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)


def test_kmeans_stream_groups_do_not_change_the_results(frames):
    """from 16 streams on the batched k-means rounds run as independent chains per group of streams: the 16-stream pipeline (two groups) equals two
    8-stream pipelines (one group each) on the same streams"""
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames
    T = 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    vb = [bgr, bgr[:, :, ::-1], bgr[:, ::-1], bgr[:, ::-1, ::-1]]; vd = [depth, depth[:, :, ::-1], depth[:, ::-1], depth[:, ::-1, ::-1]]
    gain = [1.0, 0.9, 0.8, 0.7]
    sb = np.stack([np.ascontiguousarray((vb[s % 4] * gain[s // 4]).astype(np.uint8)) for s in range(16)])
    sd = np.stack([np.ascontiguousarray(vd[s % 4]) for s in range(16)])

    def run(streams, groups=-1):
        pipe = Pipeline(len(streams), T, 640, 480, *K, 1500, 1.2, 8, 15, 5)
        if groups > 0:
            pipe.set_kmeans_groups(groups)
        for i, s in enumerate(streams):
            pipe.prime(i, sb[s, 1], sb[s, 0])
        got = []
        for step in range(2):
            lo = 2 + step * T
            pipe.process(sb[streams, lo:lo + T], sd[streams, lo:lo + T])
            got.append([np.asarray(a).reshape((len(streams), T, -1)).copy() for a in (pipe.dyna, pipe.label, pipe.mask, pipe.nkp)])      # frame k = stream * T + t
        groups = pipe.kmeans_groups()
        pipe.close()
        return got, groups

    whole, g16 = run(list(range(16)), 2)
    lo8, g8 = run(list(range(8))); hi8, _ = run(list(range(8, 16)))
    assert g16 == 2 and g8 == 1
    for step in range(2):
        for k in range(4):
            assert np.array_equal(whole[step][k][:8], lo8[step][k]) and np.array_equal(whole[step][k][8:], hi8[step][k])


@pytest.mark.timeout(1200)
def test_pipeline_in_the_shape_the_bench_runs_streaming_solver_and_ragged_step(frames):
    """48 streams x 3 frames = 144 pairs per step, the dense flow in ONE slice (flow_slices = 1; the rule by step size would cut 144 pairs into three slices of 48, below the
    80 images per launch from which the large-level solver takes over): the ONE-WAVE solver (k_sor_wave, the kernel of the bench's 170-pair slices) inside the pipeline; the
    workgroup pipeline it replaced (k_sor_stream, flow_opts_off bit 3) gives the same step.  Four sampled streams against the ORACLE: imgDyna / imgLabel / mask / keypoints / descriptors of every
    frame are equal.  Then a ragged step (sind_pipe_set_active_frames): the sampled streams stop after 0, 1, 2 and 3 frames -- their state fingerprints and state
    blobs are exactly those of the oracle-checked prefix."""
    from sindslam_amd.pipeline import Pipeline
    bgr, depth = frames                       # 6 frames
    S, T = 48, 3
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])

    def variant(s):
        b, d = bgr, depth
        if s & 1: b, d = b[:, :, ::-1], d[:, :, ::-1]
        if s & 2: b, d = b[:, ::-1], d[:, ::-1]
        g = 1.0 - 0.03 * ((s >> 2) % 12)
        return (np.clip(b.astype(np.float32) * g, 0, 255).astype(np.uint8) if g != 1.0 else np.ascontiguousarray(b)), np.ascontiguousarray(d)
    sb = np.empty((S, 6) + bgr.shape[1:], np.uint8); sd = np.empty((S, 6) + depth.shape[1:], np.uint16)
    for s in range(S):
        sb[s], sd[s] = variant(s)
    pipe = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1, flow_slices=1)
    pipe.set_state_hashing(True)
    for s in range(S):
        pipe.prime(s, sb[s, 1], sb[s, 0])
    pipe.process(sb[:, 2:5], sd[:, 2:5])
    st = pipe.stats(); assert st["sor_slices"] == 1          # 144 pairs in one slice: streaming (from 80 images per launch)
    sample = [0, 7, 22, 47]
    orb_ref = O.ORBextractor(1500, 1.2, 8, 15, 5)
    h_full = pipe.state_hashes(); assert (h_full != 0).any(axis=2).all()
    for s in sample:
        ref = O.DynaDetect(np.ascontiguousarray(sb[s, 1]), np.ascontiguousarray(sb[s, 0]), *K)
        for t in range(T):
            rd, rl = ref.detect(np.ascontiguousarray(sb[s, 2 + t]), np.ascontiguousarray(sd[s, 2 + t]))
            assert np.array_equal(pipe.dyna[s, t], rd) and np.array_equal(pipe.label[s, t], rl), (s, t)
            assert np.array_equal(pipe.mask[s, t], O.dilate15(rd))
            rk, rdesc = orb_ref.extract(O.bgr2gray(np.ascontiguousarray(sb[s, 2 + t]), swap_rb=True), pipe.mask[s, t])
            k, d = pipe.keypoints(s, t)
            assert k.tobytes() == rk.tobytes() and np.array_equal(d, rdesc), (s, t)
    blobs_full = {s: pipe.get_state(s) for s in sample}
    # the same step through k_sor_stream
    alt = Pipeline(S, T, 640, 480, *K, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1, flow_slices=1, flow_opts_off=8)
    alt.set_state_hashing(True)
    for s in range(S):
        alt.prime(s, sb[s, 1], sb[s, 0])
    alt.process(sb[:, 2:5], sd[:, 2:5])
    assert alt.stats()["sor_launches"] > 0 and np.array_equal(alt.dyna, pipe.dyna) and np.array_equal(alt.label, pipe.label) and np.array_equal(alt.state_hashes(), h_full)
    alt.close()
    # ragged step on the same inputs: stream sample[i] runs i frames
    for s in range(S):
        pipe.prime(s, sb[s, 1], sb[s, 0])
    act = np.full(S, T, np.int32)
    for i, s in enumerate(sample):
        act[s] = i
    pipe.set_active_frames(act)
    keep_dyna = pipe.dyna.copy()
    pipe.process(sb[:, 2:5], sd[:, 2:5])
    h_rag = pipe.state_hashes()
    for i, s in enumerate(sample):
        assert np.array_equal(h_rag[s, :i], h_full[s, :i]) and (h_rag[s, i:] == 0).all(), (s, i)
        assert np.array_equal(pipe.dyna[s, :i], keep_dyna[s, :i])
    assert np.array_equal(h_rag[1], h_full[1]) and np.array_equal(pipe.dyna[1], keep_dyna[1])           # the other streams are not affected
    assert np.array_equal(pipe.get_state(sample[3]), blobs_full[sample[3]])                              # 3 of 3 frames: the full step's state
    # the state after i frames continues to the same results: stream sample[2] (2 frames done) processes frame 3 in a one-stream pipeline
    s2 = sample[2]; one = Pipeline(1, 1, 640, 480, *K, 1500, 1.2, 8, 15, 5, orb_gray_rgb_order=1)
    one.prime(0, sb[s2, 3], sb[s2, 2]); one.set_state(0, pipe.get_state(s2)); one.set_state_hashing(True)
    one.process(sb[s2:s2 + 1, 4:5], sd[s2:s2 + 1, 4:5])
    assert np.array_equal(one.dyna[0, 0], keep_dyna[s2, 2]) and np.array_equal(one.state_hashes()[0, 0], h_full[s2, 2])
    # the call after a ragged step is a full step again
    for s in range(S):
        pipe.prime(s, sb[s, 1], sb[s, 0])
    pipe.process(sb[:, 2:5], sd[:, 2:5])
    assert np.array_equal(pipe.state_hashes(), h_full) and np.array_equal(pipe.dyna, keep_dyna)
    one.close(); pipe.close()


@pytest.mark.timeout(900)
def test_pipeline_group_equals_one_pipeline(frames):
    """the streams of a step cut into three independent pipelines driven concurrently (PipelineGroup, the small-step schedule of a multi-GPU rank): every output,
    state fingerprint and state blob equals the one-pipeline run; pipelined submit / flush and a ragged step included"""
    import torch
    from sindslam_amd.pipeline import Pipeline, PipelineGroup
    bgr, depth = frames
    S, T = 7, 2
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    sb = np.stack([np.roll(bgr, 3 * s, axis=2) if s % 2 else np.ascontiguousarray(bgr[:, ::-1] if s % 4 == 2 else bgr) for s in range(S)])
    sd = np.stack([np.roll(depth, 3 * s, axis=2) if s % 2 else np.ascontiguousarray(depth[:, ::-1] if s % 4 == 2 else depth) for s in range(S)])
    a = (S, T, 640, 480) + K + (1500, 1.2, 8, 15, 5)
    one = Pipeline(*a, orb_gray_rgb_order=1); grp = PipelineGroup(3, *a, orb_gray_rgb_order=1)
    assert [p.S for p in grp.pipes] == [2, 2, 3] and grp.host_info()["pipelines"] == 3
    dev = [(torch.from_numpy(np.ascontiguousarray(sb[:, 2 + i * T: 4 + i * T])).cuda(), torch.from_numpy(np.ascontiguousarray(sd[:, 2 + i * T: 4 + i * T]).view(np.int16)).cuda()) for i in range(2)]
    res = {}
    for name, p in (("one", one), ("grp", grp)):
        p.set_state_hashing(True)
        for s in range(S):
            p.prime(s, sb[s, 1], sb[s, 0])
        got = []
        assert not p.submit_dev(dev[0][0].data_ptr(), dev[0][1].data_ptr())
        p.set_active_frames(np.array([2, 1, 0, 2, 2, 1, 2], np.int32))
        assert p.submit_dev(dev[1][0].data_ptr(), dev[1][1].data_ptr())
        got.append((p.dyna.copy(), p.label.copy(), p.mask.copy(), p.nkp.copy(), p.kps.copy(), p.desc.copy(), p.state_hashes()))
        assert p.flush()
        got.append((p.dyna.copy(), p.label.copy(), p.mask.copy(), p.nkp.copy(), p.kps.copy(), p.desc.copy(), p.state_hashes()))
        res[name] = (got, [p.get_state(s) for s in range(S)])
    for step in range(2):
        for x, y in zip(res["one"][0][step], res["grp"][0][step]):
            assert np.array_equal(x, y), step
    for s in range(S):
        assert np.array_equal(res["one"][1][s], res["grp"][1][s]), s
    h = res["grp"][0][1][6]; assert (h[2] == 0).all() and (h[1, 1] == 0).all() and (h[1, 0] != 0).any()
    one.close(); grp.close()


def test_reserve_retained_is_exact_and_gives_memory_back():
    """sind_pipe_reserve_retained(n) means exactly n sets: a smaller request after a larger one frees the surplus (what the out-of-memory fallback of
    VerifiedChunks.prime relies on: the sets a failed larger request had completed must not stay allocated)"""
    import torch
    from sindslam_amd.pipeline import Pipeline
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    pipe = Pipeline(8, 2, 640, 480, *K, 1500, 1.2, 8, 15, 5)
    try:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
        pipe.reserve_retained(6); torch.cuda.synchronize(); free6 = torch.cuda.mem_get_info()[0]
        per_set = (free0 - free6) / 6
        assert per_set > 16 * 640 * 480 * 10                      # a set holds the flow, depth, plane-edge mask and normalised depth of the step's 16 frames
        pipe.reserve_retained(2); torch.cuda.synchronize(); free2 = torch.cuda.mem_get_info()[0]
        assert free2 - free6 > 3.5 * per_set                      # four of the six sets went back
        pipe.reserve_retained(2); torch.cuda.synchronize(); assert abs(torch.cuda.mem_get_info()[0] - free2) < 0.5 * per_set        # idempotent
        pipe.reserve_retained(0); torch.cuda.synchronize(); assert free0 - torch.cuda.mem_get_info()[0] < 0.5 * per_set
    finally:
        pipe.close()
