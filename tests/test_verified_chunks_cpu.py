"""CPU: the chunked-sequence driver (sindslam_amd/sequence.py VerifiedChunks: speculate -> verify -> repair) on a toy stateful detector (tests/fake_pipeline.py):
whatever the warm-up, the chunk count and the detector's memory, every owned frame equals the sequential loop -- seams that re-synchronise inside the
warm-up are only verified, the others are repaired frame by frame until the states agree, and a runner that reaches the end of its chunk makes the successor
verify again.  Two gloo ranks cover the cross-rank seam (fingerprints gathered, state blob sent on a mismatch)."""
import os

import numpy as np
import pytest

from fake_pipeline import FakePipeline, FakeSource, Toy
from sindslam_amd.sequence import lockstep_for, plan_lockstep, process_sequence


def _run(n_frames, streams, T, warmup, toy, world=1, rank=0, group=None, repair_streams=2, repair_T=3, verify=True, retain=0):
    src = FakeSource(n_frames - 1)
    stats = {}
    bgr = np.zeros((n_frames, 2, 3, 3), np.uint8); depth = np.zeros((n_frames, 2, 3), np.uint16)
    out = process_sequence(bgr, depth, {}, streams=streams, frames_per_step=T, warmup=warmup, rank=rank, world=world, group=group, repair_streams=repair_streams,
                           repair_frames_per_step=repair_T, verify=verify, stats=stats, pipeline_factory=lambda S_, T_: FakePipeline(S_, T_, toy, src), source=src, retain_frames=retain)
    return out, stats


def _check(out, toy, n_frames, owned=None):
    truth = toy.truth(n_frames - 1)
    for f in (owned if owned is not None else range(1, n_frames)):
        assert out["dyna"][f, 0, 0] == truth[f - 1], (f, int(out["dyna"][f, 0, 0]), truth[f - 1])
        assert out["label"][f, 0, 0] == truth[f - 1] // 2 and out["mask"][f, 0, 0] == 255 - truth[f - 1]
        assert out["keypoints"][f]["x"][0] == float(truth[f - 1])
    assert (out["dyna"][0] == 0).all()


def test_warmup_long_enough_verifies_without_repair():
    toy = Toy(bits=5)
    out, st = _run(101, 4, 4, 8, toy)
    assert out["owned"] == list(range(1, 101)); _check(out, toy, 101)
    assert st["seams"] == 3 and st["mismatched_seams"] == 0 and st["rounds"] == 0 and st["repair_frames"] == 0


def test_short_warmup_is_repaired_until_the_states_agree():
    toy = Toy(bits=9)                                            # memory of 9 frames against a warm-up of 4: every seam mismatches, a runner needs 5 frames
    out, st = _run(120, 4, 4, 4, toy)
    _check(out, toy, 120)
    assert st["mismatched_seams"] == 3 and st["rounds"] == 1 and st["repaired_chunks"] == 3 and st["runners_to_chunk_end"] == 0
    assert st["max_frames_to_converge"] == 5 and 15 <= st["repair_frames"] <= 18          # 3 runners x 5 frames, rounded up to whole repair steps of 3
    # without verification the same run is wrong behind every seam
    bad, _ = _run(120, 4, 4, 4, toy, verify=False)
    truth = toy.truth(119)
    assert sum(int(bad["dyna"][f, 0, 0] != truth[f - 1]) for f in range(1, 120)) >= 9


def test_no_warmup_at_all_every_seam_is_a_forced_mismatch():
    toy = Toy(bits=4)
    out, st = _run(64, 5, 3, 0, toy, repair_streams=3)
    _check(out, toy, 64)
    assert st["mismatched_seams"] == st["seams"] == 4 and st["rounds"] == 1


def test_runner_reaching_the_chunk_end_cascades_into_the_successor():
    toy = Toy(bits=60)                                           # never forgets inside a chunk: every runner runs to the end of its chunk, seam after seam
    out, st = _run(50, 4, 2, 2, toy, repair_streams=3, repair_T=4)          # all three runners at once: only the first starts from a true state -> three rounds
    _check(out, toy, 50)
    assert st["runners_to_chunk_end"] == 6 and st["rounds"] == 3 and st["runners"] == 6
    out, st = _run(50, 4, 2, 2, toy, repair_streams=1, repair_T=4)          # one at a time, in chunk order: each starts from its predecessor's new end state
    _check(out, toy, 50)
    assert st["runners_to_chunk_end"] == 3 and st["rounds"] == 1 and st["runners"] == 3


def test_mixed_seams_resets_resynchronise_some_chunks():
    toy = Toy(bits=40, reset_every=17)
    out, st = _run(200, 6, 5, 3, toy, repair_streams=2, repair_T=2)
    _check(out, toy, 200)
    assert 0 < st["mismatched_seams"] <= 5


@pytest.mark.parametrize("n,streams,T,warmup", [(7, 3, 2, 1), (33, 8, 1, 2), (12, 1, 5, 4), (90, 7, 6, 11)])
def test_ragged_plans(n, streams, T, warmup):
    toy = Toy(bits=7, reset_every=23)
    out, st = _run(n, streams, T, warmup, toy)
    assert out["owned"] == list(range(1, n)); _check(out, toy, n)


@pytest.mark.parametrize("retain", [4, 7, 16, 40])
def test_replay_of_retained_steps_then_the_repair_pipeline(retain):
    """memory of 14 frames against a warm-up of 3: a runner needs 11 frames.  With 4 or 7 retained frames it starts on the retained steps (tails only) and finishes on
    the repair pipeline; with 16 or 40 it never leaves them"""
    toy = Toy(bits=14)
    out, st = _run(150, 5, 3, 3, toy, retain=retain)
    _check(out, toy, 150)
    assert st["mismatched_seams"] == 4 and st["replay_frames"] > 0 and st["max_frames_to_converge"] == 11
    if retain >= 16:
        assert st["runners_past_replay"] == 0 and st["repair_frames"] == 0 and 40 <= st["replay_frames"] <= 4 * 12          # whole step ranges are replayed: a few frames past the one that matched
    else:
        assert st["runners_past_replay"] >= 3 and st["repair_frames"] > 0          # (the last, shorter chunk may end inside the retained frames)


def test_replay_cascade_and_resets():
    for bits, reset in [(60, 0), (40, 17), (9, 0)]:
        toy = Toy(bits=bits, reset_every=reset)
        out, st = _run(160, 6, 4, 2, toy, repair_streams=2, repair_T=3, retain=10)
        _check(out, toy, 160)


def test_lockstep_for_respects_the_step_size():
    for frames, n, T, W in [(4000, 26, 9, 16), (100, 4, 4, 8), (5, 3, 2, 1)]:
        p = lockstep_for(frames, n, T, W)
        assert p.T <= T and sum(c.last - c.first for c in p.chunks) == frames
        assert all(c.start == c.first - (W if g else 0) for g, c in enumerate(p.chunks) if c.last > c.first)


def _rank_worker(rank, world, port, q, bits, reset_every):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    toy = Toy(bits=bits, reset_every=reset_every)
    out, st = _run(140, 3 if world == 2 else 2, 4, 2, toy, world=world, rank=rank, repair_streams=2, repair_T=3, retain=6 if bits != 9 else 0)
    q.put((rank, out["owned"], [int(out["dyna"][f, 0, 0]) for f in out["owned"]], st["mismatched_seams"], st["rounds"], st["runners_to_chunk_end"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("bits,reset_every,world", [(9, 0, 2), (64, 0, 2), (30, 19, 2), (64, 0, 3), (11, 0, 3)])
def test_ranks_forced_mismatch_across_the_rank_seams(bits, reset_every, world):
    """warm-up 2 against a memory of 9 / 64 / 30 frames: every seam mismatches, the one between the ranks included (the end-state blob of rank 0's last chunk is
    sent to rank 1); with 64 bits the runners cascade through all chunks of both ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 30500 + (os.getpid() + 7 * bits + 101 * world) % 2000
    ps = [ctx.Process(target=_rank_worker, args=(r, world, port, q, bits, reset_every)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in ps)
    [p.join(60) for p in ps]
    truth = Toy(bits=bits, reset_every=reset_every).truth(139)
    owned = sorted(sum((r[1] for r in res), [])); assert owned == list(range(1, 140))
    for r in res:
        assert r[2] == [truth[f - 1] for f in r[1]], f"rank {r[0]}"
        assert r[3] >= 1 and r[4] >= 1
    if bits == 64:                  # never forgets inside a chunk: the runners cascade through every chunk of every rank (the middle rank of three both receives and sends a blob)
        assert sum(r[5] for r in res) >= 5
