"""One long RGB-D sequence on the batched pipeline ("frames of a TUM sequence shard naturally across the GPUs").

The state-free work of DynaDetect + ORB (>99 % of the bytes) needs only frames n, n-1, n-2 and depth n; the light stateful tail (k-means
warm labels, sample weights, previous high mask; reference DynaDetect.h:172-178, rolled at DynaDetect.cc:1660-1664) needs frame order.
So a sequence is cut into contiguous lock-step CHUNKS, one per pipeline stream (and `streams` chunks per rank); chunk 0 starts at frame 1
primed with frame 0 twice, exactly like the reference loop (Examples/RGB-D/rgbd_tum_noros.cc:103-107, 131-139); a later chunk starts `warmup`
frames early from an empty state (speculation).  VerifiedChunks then makes every chunk the sequential result: each chunk seam is verified by
comparing 128-bit fingerprints of the inter-frame state, and a chunk whose rebuilt state differs from its predecessor's true end state is
repaired by re-running the stateful tails of its first frames (on retained phase-A outputs) until the states agree.  process_sequence returns
the frames a rank owns; with several ranks the fingerprints travel in one small all_gather per round and the 1.2 MB state blob of a
mismatching seam between two ranks in one send / recv -- there is no other cross-rank dependency; the caller assembles the per-frame masks
with parallel.gather_sequence_masks (one all_gather of padded blocks).

process_sequence_exact is the in-order mode: phase A batched (and sharded over ranks by contiguous frame ranges), the stateful tails strictly
in frame order -- one chain per half (depth half: k-means warm labels; flow half: sample weights, previous high mask), the depth chain of step
i+1 next to the flow chain of step i -- and the state blob handed from rank r to rank r+1.  Its rate is bounded by 1 / (per-frame chain
latency) and does not grow with the number of ranks; it is the independent check the chunked results are compared with.
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


# ---------------------------------------------------------------------------------------------------------------- verified chunks
class HostFrames:
    """Frame source over host arrays bgr u8 [N, H, W, 3] / depth u16 [N, H, W]: sequence POSITION q (0 = the first frame DetectDynaArea processes) is array
    frame q + 1; positions -1 and -2 (the two priming frames of the sequential loop, rgbd_tum_noros.cc:103-107) are both frame 0; positions past the end
    repeat the last frame (lock-step padding, results unowned)."""

    def __init__(self, bgr: np.ndarray, depth: np.ndarray, device: int = 0):
        self.bgr, self.depth, self.device = bgr, depth, device

    def index(self, q):
        return np.clip(np.asarray(q) + 1, 0, len(self.bgr) - 1)

    def host_frame(self, q: int) -> np.ndarray:
        return self.bgr[int(self.index(q))]

    def device_batch(self, pos: np.ndarray):
        import torch
        idx = self.index(pos)
        b = torch.from_numpy(np.ascontiguousarray(self.bgr[idx])).to(f"cuda:{self.device}")
        d = torch.from_numpy(np.ascontiguousarray(self.depth[idx]).view(np.int16)).to(f"cuda:{self.device}")
        torch.cuda.synchronize(self.device)
        return b.data_ptr(), d.data_ptr(), (b, d)


NO_HASH = np.array([np.iinfo(np.uint64).max, np.iinfo(np.uint64).max], np.uint64)      # "no such state": never equal to a fingerprint


class VerifiedChunks:
    """One sequence on n = world * S lock-step chunks whose results EQUAL the sequential frame loop (speculate -> verify -> repair).

    Every output of a frame is a deterministic function of (input frames n, n-1, n-2, depth n, state before the frame); the state is the three images the
    reference rolls at DynaDetect.cc:1660-1664 (consumed at :374-395, :1169-1219, :1560).  Chunk g > 0 SPECULATES: it starts `warmup` frames early from an
    empty state and so reaches its first owned frame with some state; chunk g - 1 reaches the same frame with the TRUE state.  The pipeline leaves a 128-bit
    fingerprint of the rolled state per frame (sind_pipe_set_state_hashing), so after the K lock-step steps:
      verify   chunk g is the sequential result iff its state at position first_g - 1 equals the end state of chunk g - 1 (chunk 0 is the sequential loop).
      repair   otherwise a RUNNER re-processes chunk g from the true state, frame by frame on a small second pipeline, until its state equals the
               speculative state of the same frame -- from there on the speculative results are the sequential ones -- or the chunk ends.  A runner that
               reaches the end of its chunk changes the chunk's end state, and the successor is verified again (next round).  The first unverified chunk
               of a round always has a true predecessor, so the loop ends after at most n rounds; measured, a runner converges within ~13 frames.
    With several ranks (chunk g lives on rank g // S) a round costs one all_gather of 32 bytes per chunk, plus one send / recv of the 1.2 MB state blob for
    a seam between ranks that needs a runner.  There is no other cross-rank dependency: the path shards by frame.

    pipe / repair: Pipeline-like objects (S x T and R x Tr); source: positions -> inputs (HostFrames or the bench's device-resident source)."""

    def __init__(self, plan: LockstepPlan, S: int, pipe, repair, source, rank: int = 0, world: int = 1, group=None, retain_frames: int = 0):
        assert plan.n_chunks == S * world and pipe.S == S and pipe.T == plan.T
        self.plan, self.S, self.pipe, self.repair, self.src, self.rank, self.world, self.group = plan, S, pipe, repair, source, rank, world, group
        # REPLAY: the steps that hold the first `retain_frames` owned frames of the chunks after the first (processed frames warmup .. warmup + retain_frames of every
        # such chunk: the chunks run in lock-step, so these are the same few steps for all of them) keep their phase-A outputs (sind_pipe_retain_next), and a
        # runner re-runs only the stateful tails of those frames on the chunk's own stream (sind_pipe_replay) -- the state-free 99 % of a frame (dense flow, ORB
        # front, CalOccluded) is not computed again.  A runner that is still not through after the retained frames continues on the repair pipeline.
        # retain_frames < 0: every step from the first owned frame on (how long a runner needs is a property of the data -- the k-means of a chunk that started from other
        # labels can sit in another local optimum for as long as the scene stays similar: 46 frames at most on the bench's TUM-shaped stream, 129 with the Bonn
        # parameters, whole 92-frame chunks at 1280 x 720 --, and a replayed frame costs a third of a re-processed one; a retained step of 224 frames is ~0.8 GB of HBM)
        T = plan.T
        if retain_frames < 0:
            retain_frames = max(1, plan.processed - plan.warmup)
        self.retained = list(range(plan.warmup // T, min(plan.steps, (plan.warmup + retain_frames - 1) // T + 1))) if (retain_frames > 0 and plan.n_chunks > 1) else []
        self.mine = plan.chunks[rank * S:(rank + 1) * S]
        self.H = np.zeros((S, plan.processed, 2), np.uint64)             # fingerprint of the state after every processed frame of my chunks (current best chain)
        self.end_blob = {}                                               # local chunk -> end-state blob when a runner changed it
        self.stats = dict(seams=0, mismatched_seams=0, rounds=0, runners=0, repaired_chunks=0, repair_frames=0, repair_steps=0, overridden_frames=0,
                          runners_to_chunk_end=0, repair_seconds=0.0, max_frames_to_converge=0, replay_frames=0, replay_calls=0, runners_past_replay=0)

    # ---- the K lock-step steps
    def prime(self):
        for s, c in enumerate(self.mine):
            self.pipe.prime(s, self.src.host_frame(c.start - 1), self.src.host_frame(c.start - 2))
        self.pipe.set_state_hashing(True)
        if self.retained:
            self.pipe.release_retained(-1)
            while self.retained:                 # a retained step is ~4 KB of HBM and ~0.6 KB of page-locked host memory per pixel-frame: take what the machine gives, earliest steps first
                try:
                    self.pipe.reserve_retained(len(self.retained)); break
                except Exception as e:           # (SindError of the real pipeline: allocation failed)
                    if len(self.retained) == 1 or self.repair is None:
                        raise
                    self.stats["retained_steps_dropped"] = self.stats.get("retained_steps_dropped", 0) + len(self.retained) - len(self.retained) // 2
                    self.retained = self.retained[:len(self.retained) // 2]
                    # (the failed call kept the sets it had completed; the retry with the smaller count FREES the surplus -- reserve_retained(n) means exactly n -- so the
                    # depth-half objects, the repair pipeline's buffers and the input batches that are allocated next find memory)
            if hasattr(self.pipe, "set_depth_ahead"):
                # the last runners of a repair run as two chains per stream (depth half ahead of the flow half) on separate depth-half objects: switching depth-ahead on
                # and off again creates them now, outside any timed region, instead of at the first replay
                self.pipe.set_depth_ahead(True); self.pipe.set_depth_ahead(False)

    def positions(self, step: int) -> np.ndarray:
        T = self.plan.T
        return np.array([[c.start + step * T + t for t in range(T)] for c in self.mine], np.int64)

    def run_main(self, on_step=None, inputs=None, after_submit=None):
        """on_step(step, pipe): the results of `step` are in pipe.dyna / label / mask / kps.  inputs(step) -> (bgr_ptr, depth_ptr[, keep]) overrides the source
        (the bench builds its K input blocks before the clock starts).  after_submit(step, seconds): the submit call of `step` has returned."""
        import time
        K, T = self.plan.steps, self.plan.T; pending = None

        def collect(i):
            self.H[:, i * T:(i + 1) * T] = self.pipe.state_hashes()
            if on_step is not None:
                on_step(i, self.pipe)
        for i in range(K):
            inp = inputs(i) if inputs is not None else self.src.device_batch(self.positions(i))
            if i in self.retained:
                self.pipe.retain_next(i)
            t0 = time.perf_counter(); have = self.pipe.submit_dev(inp[0], inp[1]); dt = time.perf_counter() - t0
            if have:
                collect(pending)
            pending = i
            if after_submit is not None:
                after_submit(i, dt)
        t0 = time.perf_counter()
        if self.pipe.flush():
            collect(pending)
        self.flush_seconds = time.perf_counter() - t0

    # ---- verify / repair
    def _owns(self, g):
        c = self.plan.chunks[g]; return c.last > c.first

    def _hash_at(self, s, q):
        c = self.mine[s]
        return self.H[s, q - c.start] if c.start <= q < c.start + self.plan.processed else NO_HASH

    def _gather(self, local: np.ndarray) -> np.ndarray:
        """[S, k] u64 of every rank -> [world * S, k]"""
        if self.world == 1:
            return local
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.from_numpy(local.view(np.int64).copy()).to(dev); parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return np.concatenate([p.cpu().numpy() for p in parts]).view(np.uint64)

    def _exchange_blobs(self, need):
        """the end-state blob of the last chunk of rank r - 1 for every needed seam between ranks; returns {local chunk 0: blob} when this rank receives one"""
        got = {}
        if self.world == 1:
            return got
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        ops = []; recv_t = None; keep = None
        if self.rank + 1 < self.world and need[(self.rank + 1) * self.S]:
            keep = torch.from_numpy(self._end_blob(self.S - 1).copy()).to(dev); ops.append(dist.P2POp(dist.isend, keep, self.rank + 1, group=self.group))
        if self.rank > 0 and need[self.rank * self.S]:
            recv_t = torch.empty(self.pipe.get_state_bytes(), dtype=torch.uint8, device=dev); ops.append(dist.P2POp(dist.irecv, recv_t, self.rank - 1, group=self.group))
        if ops:                                  # one batch per rank: a rank in the middle both sends and receives, and the pairs must not wait for each other in a chain
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if recv_t is not None:
            got[0] = recv_t.cpu().numpy()
        return got

    def _end_blob(self, s):
        return self.end_blob[s] if s in self.end_blob else self.pipe.get_state(s)

    def verify_and_repair(self, on_frame=None, on_round=None):
        """on_frame(s, q, repair_pipe, slot, t): frame q of my chunk s has been re-processed from the true state, its results are in repair_pipe.dyna[slot, t] ...
        (called in frame order per chunk; the last call of a converged runner is the frame whose state matched).
        on_round(): end of a round, called on every rank (collective hook of the bench's mask exchange)."""
        import time
        t0 = time.perf_counter()
        S, n, P = self.S, self.plan.n_chunks, self.plan.processed
        start_h = np.stack([self._hash_at(s, c.first - 1) if (self.rank * S + s) > 0 else np.zeros(2, np.uint64) for s, c in enumerate(self.mine)])
        end_h = np.stack([self._hash_at(s, c.last - 1) for s, c in enumerate(self.mine)])
        first = True; saved = False
        while True:
            allh = self._gather(np.concatenate([start_h, end_h], axis=1))                    # [n, 4]
            need = [g > 0 and self._owns(g) and not np.array_equal(allh[g, 0:2], allh[g - 1, 2:4]) for g in range(n)]
            if first:
                self.stats["seams"] = sum(1 for g in range(1, n) if self._owns(g)); self.stats["mismatched_seams"] = sum(need); first = False
            if not any(need):
                break
            self.stats["rounds"] += 1
            if self.retained and not saved:          # the replay runs use the chunks' own streams: take every chunk's end state out first
                for s in range(S):
                    self.end_blob.setdefault(s, self.pipe.get_state(s))
                saved = True
            got = self._exchange_blobs(need)
            local = [s for s in range(S) if need[self.rank * S + s]]
            starts = {}                              # runner of chunk s: (first position to process, state blob before it)
            for s in local:
                starts[s] = (self.mine[s].first, got[0] if (s == 0 and 0 in got) else self._end_blob(s - 1))
                # the state this chunk's chain now starts from (the gathered value; a local predecessor that gets a new end state in this round re-opens the seam)
                start_h[s] = allh[self.rank * S + s - 1, 2:4]
            self.stats["runners"] += len(local)
            if self.retained:
                starts = self._replay_runners(starts, end_h, on_frame)
            if starts:
                if self.repair is None:
                    raise RuntimeError("VerifiedChunks: a chunk needs more repair than the retained frames hold and there is no repair pipeline")
                order = sorted(starts); R = self.repair.S
                for b0 in range(0, len(order), R):
                    batch = {}
                    for s in order[b0:b0 + R]:
                        q0, blob = starts[s]
                        if q0 == self.mine[s].first and s > 0 and s - 1 in self.end_blob:
                            # a runner that has not started yet takes its LOCAL predecessor's newest end state: batches run in chunk order, so an earlier batch
                            # of this round may just have made it true (saves the round that would otherwise re-open this seam)
                            blob = self.end_blob[s - 1]; start_h[s] = end_h[s - 1]
                        batch[s] = (q0, blob)
                    self._run_runners(batch, end_h, on_frame)
            if on_round is not None:
                on_round()
        self.stats["repair_seconds"] += time.perf_counter() - t0
        return self.stats

    def _take(self, s, q, hh, pipe_obj, slot, t, end_h, on_frame):
        """frame q of my chunk s has been re-processed from the true state (fingerprint hh): hand the results on, then decide -- True: the runner is done"""
        c = self.mine[s]; i = q - c.start
        if on_frame is not None:
            on_frame(s, q, pipe_obj, slot, t)
        self.stats["overridden_frames"] += 1
        if np.array_equal(hh, self.H[s, i]):                     # same state as the chain that is already there: the rest of it stands
            self.stats["max_frames_to_converge"] = max(self.stats["max_frames_to_converge"], q - c.first + 1)
            self.stats["repaired_chunks"] += 1; return True
        self.H[s, i] = hh
        if q + 1 >= c.last:                                      # the runner IS the chunk now: new end state, the successor is verified again
            end_h[s] = hh; self.end_blob[s] = pipe_obj.get_state(slot)
            self.stats["runners_to_chunk_end"] += 1; self.stats["repaired_chunks"] += 1; return True
        return False

    def _replay_runners(self, starts, end_h, on_frame):
        """runners on the chunks' own streams over the retained steps (tails only); returns the runners that are still not through: {s: (next position, state)}"""
        T, S, pipe = self.plan.T, self.S, self.pipe
        for s, (q0, blob) in starts.items():
            pipe.set_state(s, blob)
        live = {s: q0 for s, (q0, _) in starts.items()}
        for k in self.retained:
            t0 = np.zeros(S, np.int32); t1 = np.zeros(S, np.int32)
            for s, q in live.items():
                c = self.mine[s]; lo = max(q - c.start, k * T); hi = min(c.last - c.start, (k + 1) * T)
                if lo < hi:
                    assert lo == q - c.start, "a runner's frames are consecutive"
                    t0[s] = lo - k * T; t1[s] = hi - k * T
            if not (t1 > t0).any():
                continue
            pipe.replay(k, t0, t1); hh = pipe.state_hashes()
            self.stats["replay_calls"] += 1; self.stats["replay_frames"] += int((t1 - t0).sum())
            for s in list(live):
                c = self.mine[s]
                for t in range(int(t0[s]), int(t1[s])):
                    q = c.start + k * T + t
                    if self._take(s, q, hh[s, t], pipe, s, t, end_h, on_frame):
                        del live[s]; break
                    live[s] = q + 1
        left = {s: (q, pipe.get_state(s)) for s, q in live.items()}
        self.stats["runners_past_replay"] += len(left)
        return left

    def _run_runners(self, batch, end_h, on_frame):
        """full re-processing on the repair pipeline: batch = {my chunk s: (first position, state before it)}, at most repair.S of them"""
        rp, Tr = self.repair, self.repair.T
        slots = {}
        for j, (s, (q0, blob)) in enumerate(sorted(batch.items())):
            rp.prime(j, self.src.host_frame(q0 - 1), self.src.host_frame(q0 - 2)); rp.set_state(j, blob)
            slots[j] = [s, q0]
        while slots:
            pos = np.zeros((rp.S, Tr), np.int64); act = np.zeros(rp.S, np.int32)
            for j in range(rp.S):
                if j in slots:
                    s, q = slots[j]; act[j] = min(Tr, self.mine[s].last - q); pos[j] = q + np.arange(Tr)
                else:
                    pos[j] = np.arange(Tr)                                   # idle slot: any valid frames, no tail runs
            rp.set_active_frames(act)
            inp = self.src.device_batch(pos)
            rp.process_dev(inp[0], inp[1])
            hh = rp.state_hashes(); self.stats["repair_steps"] += 1; self.stats["repair_frames"] += int(act.sum())
            for j in list(slots):
                s = slots[j][0]
                for t in range(int(act[j])):
                    if self._take(s, int(pos[j, t]), hh[j, t], rp, j, t, end_h, on_frame):
                        del slots[j]; break
                    slots[j][1] = int(pos[j, t]) + 1


def lockstep_for(frames: int, n_chunks: int, frames_per_step: int, warmup: int) -> LockstepPlan:
    """the lock-step plan whose steps hold at most `frames_per_step` frames per chunk"""
    need = -(-(frames + warmup * (n_chunks - 1)) // n_chunks)
    if n_chunks > 1:
        need = max(need, warmup + 1)
    return plan_lockstep(frames, n_chunks, -(-need // max(1, frames_per_step)), warmup)


def process_sequence(bgr: np.ndarray, depth: np.ndarray, intr: dict, streams: int = 8, frames_per_step: int = 4, warmup: int = 16,
                     nfeatures: int = 1500, scale_factor: float = 1.2, nlevels: int = 8, orb_gray_rgb_order: int = 1, device: int = 0,
                     rank: int = 0, world: int = 1, want_keypoints: bool = True, group=None, repair_streams: int = 0, repair_frames_per_step: int = 4,
                     verify: bool = True, stats: dict | None = None, pipeline_factory=None, source=None, retain_frames: int = -1):
    """bgr u8 [N, H, W, 3], depth u16 [N, H, W] (host) -> dict with dyna / label / mask u8 [N, H, W] (frame 0 stays zero, like the reference's first frame)
    and, if asked, per-frame keypoint / descriptor lists, for the frames this rank owns (`owned` = sorted frame indices).  All ranks must pass the same
    sequence and parameters.  The sequence runs on streams * world lock-step chunks (VerifiedChunks): with verify (default) every owned frame equals the
    sequential loop (rgbd_tum_noros.cc:110-170) bit for bit; verify=False keeps the speculative results of the chunks (round-3 behaviour: valid masks,
    not identical near the seams).  With several ranks pass the process group (seam fingerprints and, for a mismatching seam between ranks, the state blob
    travel over it); the caller assembles the masks with parallel.gather_sequence_masks."""
    from .pipeline import Pipeline
    n, h, w, _ = bgr.shape
    plan = lockstep_for(n - 1, streams * world, frames_per_step, warmup)
    mk = pipeline_factory or (lambda S_, T_: Pipeline(S_, T_, w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], nfeatures, scale_factor, nlevels,
                                                      intr["ini_th"], intr["min_th"], orb_gray_rgb_order=orb_gray_rgb_order, device=device))
    src = source or HostFrames(bgr, depth, device)
    pipe = mk(streams, plan.T); rp = None
    out = dict(dyna=np.zeros((n, h, w), np.uint8), label=np.zeros((n, h, w), np.uint8), mask=np.zeros((n, h, w), np.uint8), owned=[],
               keypoints=[None] * n, descriptors=[None] * n)
    try:
        if verify and plan.n_chunks > 1:
            rp = mk(repair_streams or max(1, min(streams, 8)), max(1, repair_frames_per_step))
            for j in range(rp.S):
                rp.prime(j, src.host_frame(-1), src.host_frame(-2))
            rp.set_state_hashing(True)
        vc = VerifiedChunks(plan, streams, pipe, rp, src, rank, world, group, retain_frames=retain_frames if rp is not None else 0)
        vc.prime()

        def take(p, s_, t_, q):
            f = q + 1
            out["dyna"][f] = p.dyna[s_, t_]; out["label"][f] = p.label[s_, t_]; out["mask"][f] = p.mask[s_, t_]
            if want_keypoints:
                k, d = p.keypoints(s_, t_); out["keypoints"][f] = k.copy(); out["descriptors"][f] = d.copy()

        def on_step(i, p):
            for s_, c in enumerate(vc.mine):
                for t_ in range(plan.T):
                    q = c.start + i * plan.T + t_
                    if c.first <= q < c.last:
                        take(p, s_, t_, q)
        vc.run_main(on_step)
        if rp is not None:
            vc.verify_and_repair(on_frame=lambda s_, q, p, j, t_: take(p, j, t_, q))
        for c in vc.mine:
            out["owned"] += [q + 1 for q in range(c.first, c.last)]
        if stats is not None:
            stats.update(vc.stats); stats["plan"] = plan
    finally:
        pipe.close()
        if rp is not None:
            rp.close()
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
