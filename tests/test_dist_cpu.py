"""CPU: the N>1 path (stream sharding + mask gather) with two gloo ranks."""
import os

import numpy as np
import pytest


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from sindslam_amd.parallel import frame_pairs_per_second, gather_masks, shard_streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(6, rank, world)
    masks = torch.stack([torch.full((2, 8, 16), 10 * s, dtype=torch.uint8) for s in mine])      # stream id encoded in the mask
    g = gather_masks(masks)
    rate = frame_pairs_per_second(len(mine) * 2, 1.0 + rank)
    q.put((rank, mine, g[:, :, 0, 0, 0].tolist(), rate))
    dist.destroy_process_group()


def test_shard_and_gather_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4, 5]
    for r in res:
        assert r[2] == [[0, 10, 20], [30, 40, 50]]                               # every rank sees all masks in rank order
        assert abs(r[3] - 12 / 2.0) < 1e-9                                       # 12 pairs over the slowest rank's 2 s


def test_shard_streams_uneven():
    from sindslam_amd.parallel import shard_streams
    parts = [shard_streams(10, r, 4) for r in range(4)]
    assert [len(p) for p in parts] == [3, 3, 2, 2] and sum(parts, []) == list(range(10))


def _seq_worker(rank, world, port, q):
    import torch.distributed as dist
    from sindslam_amd.parallel import gather_sequence_masks
    from sindslam_amd.sequence import plan_chunks
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, streams = 24, 3                                            # 23 owned frames over 2 ranks x 3 streams: 4 4 4 4 4 3 -> 12 and 11 frames
    chunks = plan_chunks(n, streams * world, 2)[rank * streams:(rank + 1) * streams]
    owned = [f for c in chunks for f in range(c.first, c.last)]
    masks = np.zeros((n, 4, 6), np.uint8)
    for f in owned:
        masks[f] = f                                              # frame index encoded in the mask
    full = gather_sequence_masks(masks, owned)
    q.put((rank, owned, full[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_sequence_masks_gather_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_seq_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res[0][1] == list(range(1, 13)) and res[1][1] == list(range(13, 24))
    for r in res:
        assert r[2] == list(range(24))                             # frame 0 stays zero, every other frame arrived from its owner
