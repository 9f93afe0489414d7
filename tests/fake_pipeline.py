"""CPU stand-in for sindslam_amd.pipeline.Pipeline in the tests of the chunked-sequence driver (sindslam_amd/sequence.py VerifiedChunks): the same calls,
a toy stateful "detector" instead of DynaDetect.  State = one B-bit shift register per stream: frame q shifts in bit(q), so two runs of the same frames
forget their different start states after exactly B frames (the real detector's state is contracting too, only less predictably); frames with
q % reset_every == 0 set the register to a constant (instant re-synchronisation).  Outputs of a frame are a function of (state BEFORE the frame, q), like the
real outputs.  The sequential truth is `truth(q_count)`."""
from __future__ import annotations

import numpy as np

KP = np.dtype([("x", np.float32), ("y", np.float32)])


def _bit(q: int) -> int:
    return (q * 2654435761 >> 7) & 1


class Toy:
    def __init__(self, bits: int = 6, reset_every: int = 0):
        self.bits, self.reset_every = bits, reset_every

    def step(self, x: int, q: int):
        """(state before frame q) -> (outputs, state after)"""
        out = (x * 7 + q * 13 + 5) % 251
        if self.reset_every and q % self.reset_every == 0 and q > 0:
            return out, 12345 % (1 << self.bits)
        return out, (x >> 1) | (_bit(q) << (self.bits - 1))

    def truth(self, n: int):
        x, outs = 0, []
        for q in range(n):
            o, x = self.step(x, q); outs.append(o)
        return outs


class FakeSource:
    """positions in, opaque handles out; host_frame(q) is just q (the fake pipeline checks that streams see consecutive positions)"""
    def __init__(self, frames: int):
        self.frames, self.batches = frames, {}

    def host_frame(self, q: int):
        return int(q)

    def device_batch(self, pos: np.ndarray):
        h = len(self.batches) + 1; self.batches[h] = np.array(pos); return h, 0, None


class FakePipeline:
    instances = []

    def __init__(self, S: int, T: int, toy: Toy, source: FakeSource, h: int = 2, w: int = 3):
        self.S, self.T, self.toy, self.src, self.h, self.w = S, T, toy, source, h, w
        self.dyna = np.zeros((S, T, h, w), np.uint8); self.label = np.zeros_like(self.dyna); self.mask = np.zeros_like(self.dyna)
        self.x = [0] * S; self.next_q = [None] * S; self.hashing = False; self.active = None; self.pending = None
        self.hash_out = np.zeros((S, T, 2), np.uint64); self.kp = [[None] * T for _ in range(S)]
        self.frames_processed = 0; self.closed = False
        FakePipeline.instances.append(self)

    def close(self): self.closed = True
    def prime(self, s, last, lastlast): self.x[s] = 0; self.next_q[s] = int(last) + 1
    def set_state_hashing(self, on=True): self.hashing = bool(on)
    def state_hashes(self): assert self.hashing; return self.hash_out.copy()
    def get_state_bytes(self): return 16
    def get_state(self, s=0): assert self.pending is None; return np.frombuffer(np.array([self.x[s], 0], np.uint64).tobytes(), np.uint8).copy()
    def set_state(self, s, blob): self.x[s] = int(np.frombuffer(np.asarray(blob, np.uint8).tobytes(), np.uint64)[0])

    def set_active_frames(self, a):
        self.active = None if a is None else [int(v) for v in a]
        if self.active is not None:
            assert len(self.active) == self.S and all(0 <= v <= self.T for v in self.active)

    def _run(self, handle):
        pos = self.src.batches[handle]; assert pos.shape == (self.S, self.T)
        if getattr(self, "retain_tag", None) is not None:
            self.kept[self.retain_tag] = pos.copy(); self.retain_tag = None
        act = self.active or [self.T] * self.S; self.active = None
        res = dict(dyna=np.zeros_like(self.dyna), hash=np.zeros((self.S, self.T, 2), np.uint64))
        for s in range(self.S):
            for t in range(act[s]):
                q = int(pos[s, t]); qq = min(q, self.src.frames - 1)          # positions past the end repeat the last frame
                if q < self.src.frames:
                    assert self.next_q[s] == q, f"stream {s}: expected position {self.next_q[s]}, got {q}"
                    self.next_q[s] = q + 1
                o, self.x[s] = self.toy.step(self.x[s], qq)
                res["dyna"][s, t] = o; res["hash"][s, t] = (self.x[s] + 1, qq * 0 + 77)
                self.frames_processed += 1
            if act[s] < self.T:
                self.next_q[s] = None                                          # ragged step: the stream has to be primed again
        return res

    # retained steps: the fake keeps the step's positions; a replay runs the toy from the streams' current states over a frame range
    def reserve_retained(self, n): self.reserve = n; self.kept = getattr(self, "kept", {})
    def release_retained(self, tag=-1): self.kept = {} if tag < 0 else {k: v for k, v in getattr(self, "kept", {}).items() if k != tag}
    def retain_next(self, tag):
        assert tag not in self.kept and len(self.kept) < self.reserve; self.retain_tag = tag

    def replay(self, tag, first, last):
        assert self.pending is None
        pos = self.kept[tag]
        for s in range(self.S):
            for t in range(int(first[s]), int(last[s])):
                q = int(pos[s, t]); o, self.x[s] = self.toy.step(self.x[s], min(q, self.src.frames - 1))
                self.dyna[s, t] = o; self.label[s, t] = o // 2; self.mask[s, t] = 255 - o; self.hash_out[s, t] = (self.x[s] + 1, 77)
                self.kp[s][t] = np.array([(float(o), 1.0)], KP); self.frames_replayed = getattr(self, "frames_replayed", 0) + 1

    def _publish(self, res):
        self.dyna[:] = res["dyna"]; self.label[:] = res["dyna"] // 2; self.mask[:] = 255 - res["dyna"]; self.hash_out = res["hash"]
        for s in range(self.S):
            for t in range(self.T):
                self.kp[s][t] = np.array([(float(res["dyna"][s, t, 0, 0]), 1.0)], KP)

    def process_dev(self, b, d):
        assert self.pending is None; self._publish(self._run(b))

    def submit_dev(self, b, d):
        prev = self.pending; self.pending = self._run(b)          # (the tails of the fake run at submit time; results are handed out one call later, like the real one)
        if prev is not None:
            self._publish(prev)
        return prev is not None

    def flush(self):
        prev = self.pending; self.pending = None
        if prev is not None:
            self._publish(prev)
        return prev is not None

    def keypoints(self, s, t):
        return self.kp[s][t], np.zeros((1, 32), np.uint8)
