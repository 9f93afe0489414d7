"""One long RGB-D sequence on the batched pipeline ("frames of a TUM sequence shard naturally across the GPUs").

The state-free work of DynaDetect + ORB (>99 % of the bytes) needs only frames n, n-1, n-2 and depth n; the light stateful tail (k-means
warm labels, sample weights, previous high mask; reference DynaDetect.h:172-178, rolled at DynaDetect.cc:1660-1664) needs frame order.
So a sequence of N frames is cut into contiguous CHUNKS, one per pipeline stream (and `streams` chunks per rank): every chunk is an
independent stream that starts `warmup` frames before its first owned frame, so that its tail state has settled when the owned frames
begin; the outputs of the warm-up frames are dropped.  Chunk 0 starts at frame 1 primed with frame 0 twice, exactly like the reference
loop (Examples/RGB-D/rgbd_tum_noros.cc:103-107, 131-139), so its frames equal a sequential run bit for bit; later chunks deviate from a sequential run only through the state they rebuilt
in `warmup` frames -- the state steers the masks, and a rebuilt state re-synchronises with the sequential run only after ~16-24 frames
(measured: warm-up 4 or 8 leaves per-frame IoU of 0.05-0.9 over whole chunks, 16 frames >= 0.98, 20 frames mean 0.9995), hence the
default of 24; the masks of later chunks are valid but not guaranteed identical.  With several ranks, rank r owns chunks
[r * streams, (r + 1) * streams); process_sequence performs NO collective -- the caller assembles the per-frame masks of all ranks with
parallel.gather_sequence_masks (one all_gather of padded blocks).

Where the masks must EQUAL one sequential run (the parity mode, "exact"), use process_sequence_exact: the state-free phase A is still
batched (and sharded over ranks by contiguous frame ranges), the stateful tails run strictly in frame order -- one chain per half
(depth half: k-means warm labels; flow half: sample weights, previous high mask), the depth chain of step i+1 next to the flow
chain of step i -- and the state blob is handed from rank r to rank r+1 (point-to-point send / recv).  Its rate is bounded by
1 / (per-frame chain latency), not by the GPU, and does not grow with the number of ranks.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Chunk:
    first: int          # first owned frame (index into the sequence, >= 1)
    last: int           # one past the last owned frame
    start: int          # first PROCESSED frame = first - warm-up (>= 1); frames start-1 and max(start-2, 0) prime the stream

    @property
    def processed(self) -> int:
        return self.last - self.start


def plan_chunks(n_frames: int, n_chunks: int, warmup: int = 24) -> list[Chunk]:
    """Contiguous chunks over frames [1, n_frames): the first (owned % n_chunks) chunks take one frame more; every chunk after the
    first starts `warmup` frames early (never before frame 1).  Chunks may be empty when there are more chunks than frames."""
    if n_frames < 2 or n_chunks < 1 or warmup < 0:
        raise ValueError("plan_chunks: need at least 2 frames, 1 chunk and a non-negative warm-up")
    owned = n_frames - 1
    base, extra = divmod(owned, n_chunks)
    chunks, first = [], 1
    for c in range(n_chunks):
        n = base + (1 if c < extra else 0)
        chunks.append(Chunk(first, first + n, max(1, first - (warmup if c > 0 else 0))))
        first += n
    return chunks


@dataclass
class LockstepPlan:
    """A fixed-length sequence on n lock-step chunks (bench.py --workload sequence): every chunk PROCESSES exactly `processed` = steps * T
    frames, chunk 0 from the first frame on (it owns all of them, like the sequential loop), chunk g > 0 starts `warmup` frames before its
    first owned frame.  Frame numbers are positions among the processed frames of the sequence (0 = the first frame DetectDynaArea sees)."""
    frames: int                 # owned frames of the whole job (what throughput counts)
    n_chunks: int
    steps: int
    T: int                      # frames of every chunk per step
    warmup: int
    chunks: list                # Chunk(first, last, start) per chunk; a chunk past the end of the sequence owns nothing (first == last)

    @property
    def processed(self) -> int:
        return self.steps * self.T

    @property
    def processed_total(self) -> int:
        return self.processed * self.n_chunks


def plan_lockstep(frames: int, n_chunks: int, steps: int, warmup: int = 24) -> LockstepPlan:
    """Cut `frames` owned frames into n_chunks chunks that all process the same number of frames in `steps` steps: P = steps * T with the
    smallest T for which P + (n - 1) * (P - warmup) >= frames.  Chunk 0 owns [0, P), chunk g owns the next P - warmup frames; what lies
    beyond `frames` is processed (the chunks run in lock-step) but owned by nobody."""
    if frames < 1 or n_chunks < 1 or steps < 1 or warmup < 0:
        raise ValueError("plan_lockstep: need at least one frame, chunk and step and a non-negative warm-up")
    need = -(-(frames + warmup * (n_chunks - 1)) // n_chunks)            # smallest P that covers the sequence
    if n_chunks > 1:
        need = max(need, warmup + 1)                                      # a later chunk must own something
    T = -(-need // steps); P = steps * T
    chunks, first = [], 0
    for g in range(n_chunks):
        n = P if g == 0 else P - warmup
        last = min(first + n, frames); f0 = min(first, frames)
        chunks.append(Chunk(f0, last, first - (warmup if g else 0)))      # start may lie before `first` even when the chunk owns nothing: it still runs
        first += n
    return LockstepPlan(frames, n_chunks, steps, T, warmup, chunks)


def process_sequence(bgr: np.ndarray, depth: np.ndarray, intr: dict, streams: int = 8, frames_per_step: int = 4, warmup: int = 24,
                     nfeatures: int = 1500, scale_factor: float = 1.2, nlevels: int = 8, orb_gray_rgb_order: int = 1, device: int = 0,
                     rank: int = 0, world: int = 1, want_keypoints: bool = True):
    """bgr u8 [N, H, W, 3], depth u16 [N, H, W] (host) -> dict with dyna / label / mask u8 [N, H, W] (frame 0 stays zero, like the
    reference's first frame) and, if asked, per-frame keypoint / descriptor lists, for the frames this rank owns (`owned` = sorted
    frame indices).  All ranks must pass the same sequence and parameters."""
    from .pipeline import Pipeline
    n, h, w, _ = bgr.shape
    chunks = plan_chunks(n, streams * world, warmup)[rank * streams:(rank + 1) * streams]
    S, T = streams, frames_per_step
    pipe = Pipeline(S, T, w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], nfeatures, scale_factor, nlevels,
                    intr["ini_th"], intr["min_th"], orb_gray_rgb_order=orb_gray_rgb_order, device=device)
    out = dict(dyna=np.zeros((n, h, w), np.uint8), label=np.zeros((n, h, w), np.uint8), mask=np.zeros((n, h, w), np.uint8), owned=[],
               keypoints=[None] * n, descriptors=[None] * n)
    try:
        for s, c in enumerate(chunks):
            a = min(c.start, n - 1)                     # an empty chunk still needs a primed stream; it processes repeats of a valid frame
            pipe.prime(s, bgr[a - 1], bgr[max(a - 2, 0)])
        steps = max((c.processed + T - 1) // T for c in chunks) if chunks else 0
        sb = np.empty((S, T, h, w, 3), np.uint8); sd = np.empty((S, T, h, w), np.uint16)
        for step in range(steps):
            idx = np.empty((S, T), np.int64)
            for s, c in enumerate(chunks):
                for t in range(T):
                    f = c.start + step * T + t
                    idx[s, t] = f if f < c.last else -1
                    g = min(max(f if f < c.last else c.last - 1, 1), n - 1)          # past the end of a chunk: repeat its last frame, outputs dropped
                    sb[s, t] = bgr[g]; sd[s, t] = depth[g]
            pipe.process(sb, sd)
            for s, c in enumerate(chunks):
                for t in range(T):
                    f = int(idx[s, t])
                    if f < c.first:                    # -1 (padding) or a warm-up frame
                        continue
                    out["dyna"][f] = pipe.dyna[s, t]; out["label"][f] = pipe.label[s, t]; out["mask"][f] = pipe.mask[s, t]; out["owned"].append(f)
                    if want_keypoints:
                        k, d = pipe.keypoints(s, t); out["keypoints"][f] = k.copy(); out["descriptors"][f] = d.copy()
    finally:
        pipe.close()
    out["owned"].sort()
    return out


def split_frames(n_frames: int, world: int) -> list[tuple[int, int]]:
    """Contiguous frame ranges [first, last) over frames [1, n_frames) for the ranks of the exact mode (the first ranks take one more)."""
    owned = n_frames - 1
    base, extra = divmod(owned, world)
    out, first = [], 1
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((first, first + n)); first += n
    return out


def process_sequence_exact(bgr: np.ndarray, depth: np.ndarray, intr: dict, frames_per_step: int = 64, nfeatures: int = 1500, scale_factor: float = 1.2,
                           nlevels: int = 8, orb_gray_rgb_order: int = 1, device: int = 0, rank: int = 0, world: int = 1, group=None,
                           want_keypoints: bool = True, first_step: int | None = None, timing: dict | None = None):
    """In-order ("exact") run of one sequence: results equal the sequential reference loop frame for frame (rgbd_tum_noros.cc:110-170,
    state roll DynaDetect.cc:1660-1664).  Rank r owns the contiguous range split_frames(n, world)[r]; phase A of its steps is batched on
    its GPU, its tails run in frame order once the state of rank r-1 has arrived (send / recv of the state blob), and the final state goes
    to rank r+1.  Same return value as process_sequence.  first_step: length of a rank's first step, the only one whose tails do not overlap
    with a phase A (rank > 0; default frames_per_step // 4)."""
    import torch
    from .pipeline import Pipeline
    n, h, w, _ = bgr.shape
    f0, f1 = split_frames(n, world)[rank]
    out = dict(dyna=np.zeros((n, h, w), np.uint8), label=np.zeros((n, h, w), np.uint8), mask=np.zeros((n, h, w), np.uint8), owned=list(range(f0, f1)),
               keypoints=[None] * n, descriptors=[None] * n)
    dist = None
    if world > 1:
        import torch.distributed as dist
    dev = torch.device("cuda", device)
    comm_dev = dev if (world > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")

    def recv_state(nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8, device=comm_dev); dist.recv(t, src=rank - 1, group=group); return t.cpu().numpy()

    def send_state(blob):
        dist.send(torch.from_numpy(blob).to(comm_dev), dst=rank + 1, group=group)

    # step plan: [first_step (ranks > 0)] + full steps of T + one remainder step; each distinct length gets its own pipeline handle
    T = max(1, frames_per_step); L = f1 - f0; plan = []
    fs = min(L, first_step if first_step is not None else max(1, T // 4)) if rank > 0 else 0
    if fs: plan.append(fs)
    plan += [T] * ((L - fs) // T)
    if (L - fs) % T: plan.append((L - fs) % T)
    state = None; pipes = {}; t_wait = 0.0
    import time

    def pipe_for(t_len, first_frame):
        """handle with T = t_len, primed with the two frames before first_frame, carrying `state`"""
        if t_len not in pipes:
            pipes[t_len] = Pipeline(1, t_len, w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], nfeatures, scale_factor, nlevels,
                                    intr["ini_th"], intr["min_th"], orb_gray_rgb_order=orb_gray_rgb_order, device=device)
        p = pipes[t_len]
        p.prime(0, bgr[first_frame - 1], bgr[max(first_frame - 2, 0)])
        return p

    def collect(p, first_frame, t_len):
        for t in range(t_len):
            f = first_frame + t
            out["dyna"][f] = p.dyna[0, t]; out["label"][f] = p.label[0, t]; out["mask"][f] = p.mask[0, t]
            if want_keypoints:
                k, d = p.keypoints(0, t); out["keypoints"][f] = k.copy(); out["descriptors"][f] = d.copy()

    def upload(first_frame, t_len):
        b = torch.from_numpy(np.ascontiguousarray(bgr[first_frame:first_frame + t_len])).to(dev)
        d = torch.from_numpy(np.ascontiguousarray(depth[first_frame:first_frame + t_len]).view(np.int16)).to(dev)
        torch.cuda.synchronize(dev); return b, d

    try:
        f = f0; i = 0
        while i < len(plan):
            t_len = plan[i]; run = 1
            while i + run < len(plan) and plan[i + run] == t_len: run += 1            # consecutive steps of one length share a handle and pipeline
            p = pipe_for(t_len, f)
            if rank > 0 and i == 0:
                # phase A of the first step before the predecessor's state is there (no depth-ahead: both chains wait for the state)
                p.set_depth_ahead(False)
                b, d = upload(f, t_len); p.submit_dev(b.data_ptr(), d.data_ptr())
                tw = time.perf_counter(); state = recv_state(p.get_state_bytes()); t_wait += time.perf_counter() - tw
                p.set_state(0, state); p.flush(); collect(p, f, t_len); state = p.get_state(0); f += t_len; i += 1
                continue
            if state is not None: p.set_state(0, state)
            p.set_depth_ahead(True)
            prev = None
            for k in range(run):
                b, d = upload(f + k * t_len, t_len)
                if p.submit_dev(b.data_ptr(), d.data_ptr()): collect(p, prev, t_len)
                prev = f + k * t_len
            if p.flush(): collect(p, prev, t_len)
            state = p.get_state(0); f += run * t_len; i += run
        if world > 1:
            if state is None:               # a rank without frames still takes what its predecessor sends (also the LAST rank: an unmatched send would
                nb = Pipeline.state_bytes_for(w, h)          # block the predecessor forever) and forwards it
                tw = time.perf_counter(); state = recv_state(nb) if rank > 0 else np.zeros(nb, np.uint8); t_wait += time.perf_counter() - tw
            if rank + 1 < world:
                send_state(state)
    finally:
        for p in pipes.values(): p.close()
    if timing is not None: timing["state_wait_s"] = t_wait
    return out
