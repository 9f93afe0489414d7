"""One long RGB-D sequence on the batched pipeline ("frames of a TUM sequence shard naturally across the GPUs").

The state-free work of DynaDetect + ORB (>99 % of the bytes) needs only frames n, n-1, n-2 and depth n; the light stateful tail (k-means
warm labels, sample weights, previous high mask; reference DynaDetect.h:172-178, rolled at DynaDetect.cc:1660-1664) needs frame order.
So a sequence of N frames is cut into contiguous CHUNKS, one per pipeline stream (and `streams` chunks per rank): every chunk is an
independent stream that starts `warmup` frames before its first owned frame, so that its tail state has settled when the owned frames
begin; the outputs of the warm-up frames are dropped.  Chunk 0 starts at frame 1 primed with frame 0 twice, exactly like the reference
loop (Examples/RGB-D/rgbd_tum_noros.cc:103-107, 131-139), so its frames equal a sequential run bit for bit; later chunks deviate from a sequential run only through the state they rebuilt
in `warmup` frames (tests/test_sequence_gpu.py reports the mask IoU at the seams).  With several ranks, rank r owns chunks
[r * streams, (r + 1) * streams) and the per-frame masks are gathered with one all_gather per step (parallel.gather_masks).
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


def plan_chunks(n_frames: int, n_chunks: int, warmup: int = 5) -> list[Chunk]:
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


def process_sequence(bgr: np.ndarray, depth: np.ndarray, intr: dict, streams: int = 8, frames_per_step: int = 4, warmup: int = 5,
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
