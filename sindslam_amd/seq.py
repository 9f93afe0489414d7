"""One long sequence, sharded by frame, equal to the sequential loop -- ctypes mirror of the C++ driver behind sind_seq_* (include/sind_hip.h, csrc/host/seq.cpp).

The plan, the lock-step steps, the seam verification by state fingerprints, the replay / repair runners and the hand-over between ranks all run inside libsind_hip.so;
this module only passes arrays and reads the statistics.  Several ranks exchange over RCCL (a sind_comm: parallel.Comm) or over TCP (SeqNet.tcp) -- no torch.distributed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr
from .orb import KP_DTYPE
from .pipeline import PipeConfig


class SeqConfig(C.Structure):
    _fields_ = [("pipe", PipeConfig), ("frames", C.c_longlong), ("steps", C.c_int), ("frames_per_step", C.c_int), ("warmup", C.c_int), ("repair_streams", C.c_int),
                ("repair_frames_per_step", C.c_int), ("retain_frames", C.c_int), ("verify", C.c_int)]


BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong, C.POINTER(C.c_void_p))
HOOK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)

STAT_KEYS = ["seams", "mismatched_seams", "rounds", "runners", "repaired_chunks", "repair_frames", "repair_steps", "overridden_frames", "runners_to_chunk_end", "max_frames_to_converge",
             "replay_frames", "replay_calls", "runners_past_replay", "retained_steps_dropped", "repair_seconds", "flush_seconds"]


class SeqNet:
    """the exchange between the ranks of one job"""

    def __init__(self, handle, rank, world, keep=None):
        self._h, self.rank, self.world, self._keep = handle, rank, world, keep

    @classmethod
    def tcp(cls, rank: int, world: int, base_port: int, host: str | None = None):
        h = C.c_void_p()
        check(lib().sind_seq_net_tcp(rank, world, host.encode() if host else None, base_port, C.byref(h)), "sind_seq_net_tcp")
        return cls(h, rank, world)

    @classmethod
    def rccl(cls, comm):
        """comm: parallel.Comm (sind_comm on RCCL)"""
        h = C.c_void_p()
        check(lib().sind_seq_net_rccl(comm._h, C.byref(h)), "sind_seq_net_rccl")
        return cls(h, comm.rank, comm.world, keep=comm)

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_seq_net_destroy(self._h); self._h = None

    __del__ = close


class SeqJob:
    """SeqJob(frames, streams, ...) -> set_host_source / set_source, set_outputs, run() (or prime / submit(i) / flush / verify), stats()"""

    def __init__(self, frames: int, streams: int, width=640, height=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_scale=5000.0, nfeatures=1500, scale_factor=1.2, nlevels=8,
                 ini_th=15, min_th=5, orb_gray_rgb_order=0, device=0, host_threads=0, flow_max_levels=0, steps=0, frames_per_step=4, warmup=16, repair_streams=0,
                 repair_frames_per_step=4, retain_frames=-1, verify=True, net: SeqNet | None = None, flow_slices=0, flow_opts_off=0):
        pc = PipeConfig(width, height, fx, fy, cx, cy, depth_scale, nfeatures, scale_factor, nlevels, ini_th, min_th, orb_gray_rgb_order, streams, max(1, frames_per_step), device,
                        host_threads, flow_max_levels, flow_slices, flow_opts_off)
        self.cfg = SeqConfig(pc, frames, steps, frames_per_step, warmup, repair_streams, repair_frames_per_step, retain_frames, 1 if verify else 0)
        self.net = net; self.S = streams; self.w, self.h = width, height; self.cap = 2 * nfeatures + 256
        self.rank = net.rank if net else 0; self.world = net.world if net else 1
        h = C.c_void_p()
        check(lib().sind_seq_create(C.byref(self.cfg), net._h if net else None, C.byref(h)), "sind_seq_create")
        self._h = h; self._keep = []
        t = C.c_int(); k = C.c_int(); n = C.c_int()
        check(lib().sind_seq_plan(self._h, C.byref(t), C.byref(k), C.byref(n), None))
        self.T, self.steps, self.n_chunks = t.value, k.value, n.value
        fls = np.zeros((self.n_chunks, 3), np.int64)
        check(lib().sind_seq_plan(self._h, None, None, None, ptr(fls)))
        self.chunks = fls                                  # [first, last, start] per chunk (positions)
        self.mine = fls[self.rank * streams:(self.rank + 1) * streams]

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_seq_destroy(self._h); self._h = None

    __del__ = close

    def set_host_source(self, bgr: np.ndarray, depth: np.ndarray):
        assert bgr.dtype == np.uint8 and depth.dtype == np.uint16 and bgr.flags.c_contiguous and depth.flags.c_contiguous and bgr.shape[1:3] == (self.h, self.w)
        self._keep += [bgr, depth]
        check(lib().sind_seq_set_host_source(self._h, ptr(bgr), ptr(depth), C.c_longlong(len(bgr))), "sind_seq_set_host_source")

    def set_source(self, batch, frame):
        """batch(positions int64 [count]) -> (bgr device pointer, depth device pointer) of the frames at these positions, [count][H][W][3] / [count][H][W], valid until the
        next call; frame(position) -> host uint8 [H][W][3] array (kept alive by the callee until the next call)"""
        def _b(user, pos, count, ob, od):
            try:
                b, d = batch(np.ctypeslib.as_array(pos, shape=(count,)).copy())
                ob[0] = b; od[0] = d; return 0
            except Exception as e:           # an exception must not cross the C frames
                print("sind_seq batch source:", repr(e)); return -1

        def _f(user, q, out):
            try:
                a = frame(int(q)); self._frame_keep = a; out[0] = a.ctypes.data; return 0
            except Exception as e:
                print("sind_seq frame source:", repr(e)); return -1
        self._cb = (BATCH_FN(_b), FRAME_FN(_f))
        check(lib().sind_seq_set_source(self._h, self._cb[0], self._cb[1], None), "sind_seq_set_source")

    def set_outputs(self, n_frames: int, dyna=None, label=None, mask=None, kps=None, nkp=None, desc=None):
        self._keep += [dyna, label, mask, kps, nkp, desc]
        cap = 0 if kps is None and desc is None else (kps.shape[1] if kps is not None else desc.shape[1])
        check(lib().sind_seq_set_outputs(self._h, C.c_longlong(n_frames), ptr(dyna), ptr(label), ptr(mask), ptr(kps), cap, ptr(nkp), ptr(desc)), "sind_seq_set_outputs")

    def set_hooks(self, on_step=None, on_round=None):
        def wrap(fn):
            def _h(user, i):
                try:
                    fn(int(i)); return 0
                except Exception as e:
                    print("sind_seq hook:", repr(e)); return -1
            return HOOK_FN(_h) if fn else C.cast(None, HOOK_FN)
        self._hooks = (wrap(on_step), wrap(on_round))
        check(lib().sind_seq_set_hooks(self._h, self._hooks[0], self._hooks[1], None), "sind_seq_set_hooks")

    def prime(self): check(lib().sind_seq_prime(self._h), "sind_seq_prime")
    def warm(self, steps: int): check(lib().sind_seq_warm(self._h, int(steps)), "sind_seq_warm")
    def set_emit_main(self, on: bool): check(lib().sind_seq_set_emit_main(self._h, 1 if on else 0), "sind_seq_set_emit_main")
    def submit(self, step: int): check(lib().sind_seq_submit(self._h, int(step)), "sind_seq_submit")
    def flush(self): check(lib().sind_seq_flush(self._h), "sind_seq_flush")
    def verify(self): check(lib().sind_seq_verify(self._h), "sind_seq_verify")
    def run(self): check(lib().sind_seq_run(self._h), "sind_seq_run")

    def stats(self) -> dict:
        v = np.zeros(16); check(lib().sind_seq_stats(self._h, ptr(v)))
        return {k: (float(x) if k.endswith("seconds") else int(x)) for k, x in zip(STAT_KEYS, v)}

    def owned_frames(self) -> list:
        return sorted(int(q) + 1 for c in self.mine for q in range(int(c[0]), int(c[1])))

    def pipeline_handle(self):
        lib().sind_seq_pipeline.restype = C.c_void_p
        return C.c_void_p(lib().sind_seq_pipeline(self._h))

    def pipeline_view(self):
        """the main pipeline as a Pipeline object WITHOUT ownership (statistics, settings): stats(), grow_share(), kmeans_groups(), host_info(), set_chain_max_streams() ..."""
        from .pipeline import Pipeline
        v = Pipeline.__new__(Pipeline); v.S, v.T, v.w, v.h, v.cap = self.S, self.T, self.w, self.h, self.cap
        v._h = self.pipeline_handle(); v._owned = False
        return v

    def step_masks(self) -> np.ndarray:
        """the dyna array [S][T][H][W] of the last delivered step (page-locked memory of the library: a view, copy what must outlive the next step)"""
        p = C.c_void_p(); check(lib().sind_seq_step_outputs(self._h, C.byref(p), None, None))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.S, self.T, self.h, self.w))


def run_sequence(bgr: np.ndarray, depth: np.ndarray, intr: dict, streams: int = 8, frames_per_step: int = 4, warmup: int = 16, nfeatures: int = 1500, scale_factor: float = 1.2,
                 nlevels: int = 8, orb_gray_rgb_order: int = 1, device: int = 0, net: SeqNet | None = None, want_keypoints: bool = True, repair_streams: int = 0,
                 repair_frames_per_step: int = 4, verify: bool = True, stats: dict | None = None, retain_frames: int = -1, flow_max_levels: int = 0):
    """bgr u8 [N, H, W, 3], depth u16 [N, H, W] (host) -> dict with dyna / label / mask u8 [N, H, W] (frame 0 stays zero, like the reference's first frame) and, if asked,
    per-frame keypoint / descriptor lists, for the frames this rank owns (`owned`).  All ranks pass the same sequence and parameters; with verify (default) every owned frame
    equals the sequential loop (rgbd_tum_noros.cc:110-170) bit for bit."""
    n, h, w, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr); depth = np.ascontiguousarray(depth)
    job = SeqJob(n - 1, streams, w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], nfeatures, scale_factor, nlevels, intr["ini_th"], intr["min_th"],
                 orb_gray_rgb_order=orb_gray_rgb_order, device=device, flow_max_levels=flow_max_levels, frames_per_step=frames_per_step, warmup=warmup, repair_streams=repair_streams,
                 repair_frames_per_step=repair_frames_per_step, retain_frames=retain_frames, verify=verify, net=net)
    try:
        out = dict(dyna=np.zeros((n, h, w), np.uint8), label=np.zeros((n, h, w), np.uint8), mask=np.zeros((n, h, w), np.uint8), keypoints=[None] * n, descriptors=[None] * n)
        kps = nkp = desc = None
        if want_keypoints:
            kps = np.zeros((n, job.cap), KP_DTYPE); nkp = np.zeros(n, np.int32); desc = np.zeros((n, job.cap, 32), np.uint8)
        job.set_host_source(bgr, depth)
        job.set_outputs(n, out["dyna"], out["label"], out["mask"], kps, nkp, desc)
        job.run()
        out["owned"] = job.owned_frames()
        if want_keypoints:
            for f in out["owned"]:
                out["keypoints"][f] = kps[f, :nkp[f]].copy(); out["descriptors"][f] = desc[f, :nkp[f]].copy()
        if stats is not None:
            stats.update(job.stats()); stats["plan_T"] = job.T; stats["plan_steps"] = job.steps; stats["chunks"] = job.chunks.tolist()
        return out
    finally:
        job.close()
