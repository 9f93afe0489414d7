"""Multi-GPU layer: streams shard across ranks (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The hot path itself has no cross-rank dependency -- a stream is a self-contained sequence --
so the only collective is the gather of the per-frame dynamic masks that the north star asks for."""
from __future__ import annotations

import numpy as np


def shard_streams(total_streams: int, rank: int, world: int):
    """Contiguous block of stream ids for `rank`; the first (total % world) ranks take one extra."""
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def gather_masks(local_masks, group=None, out=None):
    """local_masks: torch u8 tensor [S_local, T, H, W] (equal shape on every rank) -> [world, S_local, T, H, W] on every rank.
    out: optional preallocated result tensor (reused every step by the bench)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(local_masks.shape), dtype=local_masks.dtype, device=local_masks.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(local_masks) for _ in range(world)]
        dist.all_gather(parts, local_masks.contiguous(), group=group)
        for i, p in enumerate(parts):
            out[i] = p
    else:
        dist.all_gather_into_tensor(out, local_masks.contiguous(), group=group)
    return out


def frame_pairs_per_second(local_pairs: int, local_seconds: float, group=None):
    """whole-job throughput: all ranks' units over the slowest rank's time"""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([local_seconds], dtype=torch.float64, device=dev); n = torch.tensor([local_pairs], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group); dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return float(n.item() / t.item())


def gather_sequence_masks(masks, owned, group=None):
    """One long sequence sharded by frame (sequence.process_sequence): every rank passes its u8 array [N, H, W] (numpy or torch; only the
    frames in `owned`, a contiguous ascending range, are meaningful) -> the same array with the owned frames of ALL ranks filled in, on every
    rank.  The ranks own different numbers of frames, so each contributes a block padded to the largest count (one all_gather of the
    counts, one of the blocks -- the RCCL gather of the per-frame dynamic masks of the north star)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group); backend = dist.get_backend(group)
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.as_tensor(masks)
    first = int(owned[0]) if len(owned) else 0; count = len(owned)
    if count and list(owned) != list(range(first, first + count)):
        raise ValueError("gather_sequence_masks: the owned frames of a rank must be one contiguous range")
    meta = torch.tensor([first, count], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    maxc = max(int(m[1]) for m in metas)
    block = torch.zeros((max(maxc, 1),) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    if count:
        block[:count] = t[first:first + count].to(dev)
    blocks = gather_masks(block, group=group)
    out = t.clone()
    for r in range(world):
        f, c = int(metas[r][0]), int(metas[r][1])
        if c:
            out[f:f + c] = blocks[r][:c].to(out.device)
    return out.numpy() if isinstance(masks, np.ndarray) else out


class Comm:
    """The C-ABI collective (include/sind_hip.h sind_comm_*): RCCL all-gather of byte blocks without torch.distributed -- what a C++ caller of
    include/DynaDetect.h uses.  `uid` = the 128 bytes of Comm.unique_id() made on rank 0 and passed to the other ranks by the application."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int = 0):
        import ctypes as C
        from ._lib import check, lib
        self._C, self._lib, self._check = C, lib(), check
        self.rank, self.world, self.device = rank, world, device
        h = C.c_void_p(); buf = (C.c_char * 128).from_buffer_copy(uid)
        check(self._lib.sind_comm_create(buf, rank, world, device, C.byref(h)), "sind_comm_create")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from ._lib import check, lib
        buf = (C.c_char * 128)()
        check(lib().sind_comm_unique_id(buf, 128), "sind_comm_unique_id")
        return bytes(buf)

    def allgather(self, local: np.ndarray, want_host: bool = True):
        """local: contiguous u8 array (host) -> (device tensor [world, ...] as torch u8, host copy or None)"""
        import torch
        C = self._C
        local = np.ascontiguousarray(local, np.uint8)
        out = torch.empty((self.world,) + local.shape, dtype=torch.uint8, device=f"cuda:{self.device}")
        host = np.empty((self.world,) + local.shape, np.uint8) if want_host else None
        self._check(self._lib.sind_comm_allgather_u8(self._h, local.ctypes.data_as(C.c_void_p), C.c_size_t(local.size), C.c_void_p(out.data_ptr()),
                                                     host.ctypes.data_as(C.c_void_p) if want_host else None), "sind_comm_allgather_u8")
        return out, host

    def gather_pipeline_masks(self, pipe, out=None):
        """one step's dynamic masks of every rank: torch u8 [world, S, T, H, W] on the device"""
        import torch
        C = self._C
        if out is None:
            out = torch.empty((self.world,) + pipe.dyna.shape, dtype=torch.uint8, device=f"cuda:{self.device}")
        if hasattr(pipe, "_h"):
            self._check(self._lib.sind_pipe_gather_masks(pipe._h, self._h, pipe.dyna.ctypes.data_as(C.c_void_p), C.c_void_p(out.data_ptr()), None), "sind_pipe_gather_masks")
        else:                               # a PipelineGroup: its members write into one contiguous mask array
            self._check(self._lib.sind_comm_allgather_u8(self._h, pipe.dyna.ctypes.data_as(C.c_void_p), C.c_size_t(pipe.dyna.size), C.c_void_p(out.data_ptr()), None), "sind_comm_allgather_u8")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sind_comm_destroy(self._h); self._h = None

    __del__ = close
