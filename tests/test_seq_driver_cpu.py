"""CPU: the C++ chunked-sequence driver (sindslam_amd/csrc/host/seq.cpp: speculate -> verify -> replay / repair, hand-over between ranks) on a toy stateful detector
(tests/cpp/seq_fake.cpp, the C++ twin of tests/fake_pipeline.py): whatever the warm-up, the chunk count and the detector's memory, every owned frame equals the sequential
loop.  The same scenarios as tests/test_verified_chunks_cpu.py runs on the Python driver, plus the plans of both drivers side by side; several ranks are separate
PROCESSES that exchange fingerprints and state blobs over loopback TCP (sind_seq_net_tcp's transport) -- no torch.distributed anywhere."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

from fake_pipeline import Toy

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        import cpp_shim
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libseq_fake_{os.getpid()}.so")
        _LIB = C.CDLL(cpp_shim.build_seq_fake(path))
    return _LIB


STAT_KEYS = ["seams", "mismatched_seams", "rounds", "runners", "repaired_chunks", "repair_frames", "repair_steps", "overridden_frames", "runners_to_chunk_end", "max_frames_to_converge",
             "replay_frames", "replay_calls", "runners_past_replay", "retained_steps_dropped", "repair_seconds", "flush_seconds"]


def run(n_frames, streams, T, warmup, bits, reset_every=0, world=1, rank=0, port=0, repair_streams=2, repair_T=3, verify=True, retain=0, so=None):
    L = C.CDLL(so) if so else lib()
    dyna = np.zeros(n_frames, np.uint8); label = np.zeros(n_frames, np.uint8); mask = np.zeros(n_frames, np.uint8); kpx = np.zeros(n_frames, np.float32); owned = np.zeros(n_frames, np.uint8)
    st = np.zeros(16); cnt = np.zeros(3, np.int64); err = C.create_string_buffer(512)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.seq_fake_run(C.c_longlong(n_frames), streams, T, warmup, bits, reset_every, world, rank, port, repair_streams, repair_T, 1 if verify else 0, retain,
                        p(dyna), p(label), p(mask), p(kpx), p(owned), p(st), p(cnt), err, 512)
    assert rc == 0, err.value.decode()
    stats = {k: (float(v) if k.endswith("seconds") else int(v)) for k, v in zip(STAT_KEYS, st)}
    return dict(dyna=dyna, label=label, mask=mask, kpx=kpx, owned=[int(f) for f in np.nonzero(owned)[0]]), stats, cnt


def check(out, toy, n_frames, owned=None):
    truth = toy.truth(n_frames - 1)
    for f in (owned if owned is not None else range(1, n_frames)):
        assert out["dyna"][f] == truth[f - 1], (f, int(out["dyna"][f]), truth[f - 1])
        assert out["label"][f] == truth[f - 1] // 2 and out["mask"][f] == 255 - truth[f - 1] and out["kpx"][f] == float(truth[f - 1])
    assert out["dyna"][0] == 0


def test_warmup_long_enough_verifies_without_repair():
    out, st, _ = run(101, 4, 4, 8, 5)
    assert out["owned"] == list(range(1, 101)); check(out, Toy(bits=5), 101)
    assert st["seams"] == 3 and st["mismatched_seams"] == 0 and st["rounds"] == 0 and st["repair_frames"] == 0


def test_short_warmup_is_repaired_until_the_states_agree():
    out, st, _ = run(120, 4, 4, 4, 9)                               # memory of 9 frames against a warm-up of 4: every seam mismatches, a runner needs 5 frames
    check(out, Toy(bits=9), 120)
    assert st["mismatched_seams"] == 3 and st["rounds"] == 1 and st["repaired_chunks"] == 3 and st["runners_to_chunk_end"] == 0
    assert st["max_frames_to_converge"] == 5 and 15 <= st["repair_frames"] <= 18
    bad, _, _ = run(120, 4, 4, 4, 9, verify=False)                  # without verification the same run is wrong behind every seam
    truth = Toy(bits=9).truth(119)
    assert sum(int(bad["dyna"][f] != truth[f - 1]) for f in range(1, 120)) >= 9


def test_no_warmup_at_all_every_seam_is_a_forced_mismatch():
    out, st, _ = run(64, 5, 3, 0, 4, repair_streams=3)
    check(out, Toy(bits=4), 64)
    assert st["mismatched_seams"] == st["seams"] == 4 and st["rounds"] == 1


def test_runner_reaching_the_chunk_end_cascades_into_the_successor():
    toy = Toy(bits=60)                                               # never forgets inside a chunk: every runner runs to the end of its chunk, seam after seam
    out, st, _ = run(50, 4, 2, 2, 60, repair_streams=3, repair_T=4)  # all three runners at once: only the first starts from a true state -> three rounds
    check(out, toy, 50)
    assert st["runners_to_chunk_end"] == 6 and st["rounds"] == 3 and st["runners"] == 6
    out, st, _ = run(50, 4, 2, 2, 60, repair_streams=1, repair_T=4)  # one at a time, in chunk order: each starts from its predecessor's new end state
    check(out, toy, 50)
    assert st["runners_to_chunk_end"] == 3 and st["rounds"] == 1 and st["runners"] == 3


def test_mixed_seams_resets_resynchronise_some_chunks():
    out, st, _ = run(200, 6, 5, 3, 40, reset_every=17, repair_streams=2, repair_T=2)
    check(out, Toy(bits=40, reset_every=17), 200)
    assert 0 < st["mismatched_seams"] <= 5


@pytest.mark.parametrize("n,streams,T,warmup", [(7, 3, 2, 1), (33, 8, 1, 2), (12, 1, 5, 4), (90, 7, 6, 11)])
def test_ragged_plans(n, streams, T, warmup):
    out, st, _ = run(n, streams, T, warmup, 7, reset_every=23)
    assert out["owned"] == list(range(1, n)); check(out, Toy(bits=7, reset_every=23), n)


@pytest.mark.parametrize("retain", [4, 7, 16, 40])
def test_replay_of_retained_steps_then_the_repair_pipeline(retain):
    """memory of 14 frames against a warm-up of 3: a runner needs 11 frames.  With 4 or 7 retained frames it starts on the retained steps (tails only) and finishes on
    the repair pipeline; with 16 or 40 it never leaves them"""
    out, st, cnt = run(150, 5, 3, 3, 14, retain=retain)
    check(out, Toy(bits=14), 150)
    assert st["mismatched_seams"] == 4 and st["replay_frames"] > 0 and st["max_frames_to_converge"] == 11 and cnt[2] == st["replay_frames"]
    if retain >= 16:
        assert st["runners_past_replay"] == 0 and st["repair_frames"] == 0 and 40 <= st["replay_frames"] <= 4 * 12
    else:
        assert st["runners_past_replay"] >= 3 and st["repair_frames"] > 0


def test_replay_cascade_and_resets():
    for bits, reset in [(60, 0), (40, 17), (9, 0)]:
        out, st, _ = run(160, 6, 4, 2, bits, reset_every=reset, repair_streams=2, repair_T=3, retain=10)
        check(out, Toy(bits=bits, reset_every=reset), 160)


def test_same_statistics_as_the_python_driver():
    """the C++ driver is the Python driver (sindslam_amd/sequence.py VerifiedChunks) statement for statement: same plan, same runners, same counts"""
    from fake_pipeline import FakePipeline, FakeSource
    from sindslam_amd.sequence import process_sequence
    for (n, S, T, W, bits, reset, R, Tr, retain) in [(150, 5, 3, 3, 14, 0, 2, 3, 7), (160, 6, 4, 2, 40, 17, 2, 3, 10), (50, 4, 2, 2, 60, 0, 3, 4, 0), (200, 6, 5, 3, 40, 17, 2, 2, 0)]:
        toy = Toy(bits=bits, reset_every=reset); src = FakeSource(n - 1); pst = {}
        process_sequence(np.zeros((n, 2, 3, 3), np.uint8), np.zeros((n, 2, 3), np.uint16), {}, streams=S, frames_per_step=T, warmup=W, repair_streams=R, repair_frames_per_step=Tr,
                         stats=pst, pipeline_factory=lambda S_, T_: FakePipeline(S_, T_, toy, src), source=src, retain_frames=retain)
        _, cst, _ = run(n, S, T, W, bits, reset_every=reset, repair_streams=R, repair_T=Tr, retain=retain)
        for k in STAT_KEYS[:13]:
            assert cst[k] == pst[k], (k, cst[k], pst[k], (n, S, T, W, bits))


def test_plans_equal_the_python_plans():
    from sindslam_amd.sequence import lockstep_for, plan_lockstep
    for frames, n, T, W in [(4000, 26, 9, 16), (100, 4, 4, 8), (5, 3, 2, 1), (829, 14, 13, 16), (3999, 64, 4, 16)]:
        for steps in (0, 20):
            p = plan_lockstep(frames, n, steps, W) if steps else lockstep_for(frames, n, T, W)
            t = C.c_int(); st = C.c_int(); fls = np.zeros(3 * n, np.int64)
            assert lib().seq_fake_plan(C.c_longlong(frames), n, T, steps, W, C.byref(t), C.byref(st), fls.ctypes.data_as(C.c_void_p)) == 0
            assert (t.value, st.value) == (p.T, p.steps) and fls.reshape(n, 3).tolist() == [[c.first, c.last, c.start] for c in p.chunks]


def _rank_worker(so, rank, world, port, q, bits, reset_every):
    out, st, _ = run(140, 3 if world == 2 else 2, 4, 2, bits, reset_every=reset_every, world=world, rank=rank, port=port, repair_streams=2, repair_T=3, retain=6 if bits != 9 else 0, so=so)
    q.put((rank, out["owned"], [int(out["dyna"][f]) for f in out["owned"]], st["mismatched_seams"], st["rounds"], st["runners_to_chunk_end"]))


@pytest.mark.parametrize("bits,reset_every,world", [(9, 0, 2), (64, 0, 2), (30, 19, 2), (64, 0, 3), (11, 0, 3)])
def test_ranks_forced_mismatch_across_the_rank_seams(bits, reset_every, world):
    """warm-up 2 against a memory of 9 / 64 / 30 frames: every seam mismatches, the one between the ranks included (the end-state blob of rank 0's last chunk is handed to
    rank 1 over TCP); with 64 bits the runners cascade through all chunks of all ranks (the middle rank of three both receives and sends a blob)"""
    import cpp_shim
    so = cpp_shim.build_seq_fake(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libseq_fake_mp_{os.getpid()}.so"))
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 31000 + (os.getpid() * 7 + 13 * bits + 101 * world) % 20000
    ps = [ctx.Process(target=_rank_worker, args=(so, r, world, port, q, bits, reset_every)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in ps)
    [p.join(60) for p in ps]
    truth = Toy(bits=bits, reset_every=reset_every).truth(139)
    owned = sorted(sum((r[1] for r in res), [])); assert owned == list(range(1, 140))
    for r in res:
        assert r[2] == [truth[f - 1] for f in r[1]], f"rank {r[0]}"
        assert r[3] >= 1 and r[4] >= 1
    if bits == 64:
        assert sum(r[5] for r in res) >= 5
