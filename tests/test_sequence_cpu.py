"""CPU: chunk planning of one long sequence over pipeline streams / ranks (sindslam_amd/sequence.py)."""
import pytest

from sindslam_amd.sequence import plan_chunks, plan_lockstep


@pytest.mark.parametrize("n,chunks,warm", [(4000, 64, 5), (823, 8, 5), (10, 4, 3), (3, 8, 5), (100, 1, 5), (17, 16, 0)])
def test_chunks_partition_the_sequence(n, chunks, warm):
    cs = plan_chunks(n, chunks, warm)
    assert len(cs) == chunks and cs[0].first == 1 and cs[0].start == 1 and cs[-1].last == n
    for a, b in zip(cs, cs[1:]):
        assert a.last == b.first                              # contiguous, no frame twice
    sizes = [c.last - c.first for c in cs]
    assert sum(sizes) == n - 1 and max(sizes) - min(sizes) <= 1
    for c in cs[1:]:
        assert c.start == max(1, c.first - warm) and c.processed == c.last - c.start


def test_ranks_take_disjoint_blocks_of_chunks():
    cs = plan_chunks(4000, 8 * 64, 5)                          # 8 ranks x 64 streams
    owned = [set(range(c.first, c.last)) for c in cs]
    for r in range(8):
        mine = set().union(*owned[r * 64:(r + 1) * 64])
        assert 64 * 7 <= len(mine) <= 64 * 8                   # 3999 frames in 512 chunks of 7 or 8: every stream of every rank runs the same number of steps
    assert set().union(*owned) == set(range(1, 4000))


def test_bad_arguments():
    with pytest.raises(ValueError):
        plan_chunks(1, 4)
    with pytest.raises(ValueError):
        plan_chunks(10, 0)


@pytest.mark.parametrize("frames,chunks,steps,warm", [(4000, 32, 20, 24), (4000, 26, 20, 24), (200, 4, 4, 24), (4000, 1, 20, 24), (100, 8, 3, 24), (50, 16, 2, 24), (4000, 256, 20, 24)])
def test_lockstep_plan_covers_the_sequence_once(frames, chunks, steps, warm):
    p = plan_lockstep(frames, chunks, steps, warm)
    assert p.steps == steps and p.processed == steps * p.T and len(p.chunks) == chunks
    c0 = p.chunks[0]
    assert c0.start == 0 and c0.first == 0                                  # chunk 0 is the sequential loop itself: no warm-up, owns what it processes
    owned = []
    for g, c in enumerate(p.chunks):
        owned += list(range(c.first, c.last))
        if c.last > c.first:
            assert c.start <= c.first and c.last <= c.start + p.processed   # owned frames lie inside the processed window
            assert g == 0 or c.first - c.start == warm                      # every later chunk rebuilds its state in exactly `warm` frames
    assert owned == list(range(frames))                                     # every frame owned exactly once, in order
    # the smallest T that covers the sequence: one frame less per step would not
    if p.T > 1:
        P = steps * (p.T - 1)
        assert P + (chunks - 1) * max(P - warm, 0) < frames or (chunks > 1 and P <= warm)


def test_lockstep_bad_arguments():
    for bad in [(0, 4, 2), (10, 0, 2), (10, 2, 0)]:
        with pytest.raises(ValueError):
            plan_lockstep(*bad)


def _exact_empty_worker(rank, world, port, q):
    """process_sequence_exact with more ranks than frames: every rank is empty, the state blob still travels 0 -> 1 -> 2 and nobody blocks"""
    import os
    import numpy as np
    import torch.distributed as dist
    from sindslam_amd.sequence import process_sequence_exact
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bgr = np.zeros((1, 64, 64, 3), np.uint8); depth = np.zeros((1, 64, 64), np.uint16)
    out = process_sequence_exact(bgr, depth, dict(fx=1, fy=1, cx=0, cy=0, depth_factor=5000, ini_th=15, min_th=5), rank=rank, world=world)
    q.put((rank, out["owned"]))
    dist.destroy_process_group()


def test_exact_mode_ranks_without_frames_do_not_block():
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_exact_empty_worker, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res == [(0, []), (1, []), (2, [])] and all(p.exitcode == 0 for p in ps)
