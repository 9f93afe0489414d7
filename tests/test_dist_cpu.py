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
