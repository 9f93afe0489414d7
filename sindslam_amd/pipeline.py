"""Batched multi-stream pipeline (sind_pipe_* in include/sind_hip.h): S streams x T frames per step through
DynaDetect + 15x15 dilation + ORBextractor, the frame-loop body of the reference's rgbd_tum_noros.cc."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr
from .orb import KP_DTYPE


class PipeConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("depth_scale", C.c_float), ("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("orb_gray_rgb_order", C.c_int), ("streams", C.c_int),
                ("frames_per_step", C.c_int), ("device", C.c_int), ("host_threads", C.c_int), ("flow_max_levels", C.c_int), ("flow_slices", C.c_int), ("flow_opts_off", C.c_int)]


class Pipeline:
    def __init__(self, streams: int, frames_per_step: int, width=640, height=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_scale=5000.0,
                 nfeatures=1500, scale_factor=1.2, nlevels=8, ini_th=15, min_th=5, orb_gray_rgb_order=0, device=0, host_threads=0, flow_max_levels=0, flow_slices=0, flow_opts_off=0, _out=None):
        self.S, self.T, self.w, self.h = streams, frames_per_step, width, height
        self.cap = 2 * nfeatures + 256
        cfg = PipeConfig(width, height, fx, fy, cx, cy, depth_scale, nfeatures, scale_factor, nlevels, ini_th, min_th, orb_gray_rgb_order,
                         streams, frames_per_step, device, host_threads, flow_max_levels, flow_slices, flow_opts_off)
        h = C.c_void_p()
        check(lib().sind_pipe_create(C.byref(cfg), C.byref(h)), "sind_pipe_create")
        self._h = h
        B = streams * frames_per_step
        # outputs live in page-locked memory when torch is there (plumbing only): the multi-GPU gather uploads the masks every step
        shape = (streams, frames_per_step, height, width)
        self.dyna_pinned = None
        if _out is not None:                 # a PipelineGroup hands every member its slice of the group's output arrays
            self.dyna, self.label, self.mask, self.kps, self.nkp, self.desc = _out
            return
        try:
            import torch
            if torch.cuda.is_available():
                self.dyna_pinned = torch.zeros(shape, dtype=torch.uint8).pin_memory()
        except ImportError:
            pass
        self.dyna = self.dyna_pinned.numpy() if self.dyna_pinned is not None else np.zeros(shape, np.uint8)
        self.label = np.zeros(shape, np.uint8); self.mask = np.zeros(shape, np.uint8)
        self.kps = np.zeros((B, self.cap), KP_DTYPE); self.nkp = np.zeros(B, np.int32); self.desc = np.zeros((B, self.cap, 32), np.uint8)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):        # (a view of a pipeline that a sind_seq owns -- seq.SeqJob.pipeline_view -- never destroys it)
                lib().sind_pipe_destroy(self._h)
            self._h = None

    __del__ = close

    def prime(self, stream: int, bgr_last: np.ndarray, bgr_lastlast: np.ndarray):
        check(lib().sind_pipe_prime(self._h, stream, ptr(np.ascontiguousarray(bgr_last)), ptr(np.ascontiguousarray(bgr_lastlast))), "sind_pipe_prime")

    def process(self, bgr: np.ndarray, depth: np.ndarray):
        """bgr u8 [S, T, H, W, 3], depth u16 [S, T, H, W] (host) -> results in self.dyna / label / mask / kps / nkp / desc"""
        bgr = np.ascontiguousarray(bgr, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        assert bgr.shape == (self.S, self.T, self.h, self.w, 3) and depth.shape == (self.S, self.T, self.h, self.w)
        check(lib().sind_pipe_process(self._h, ptr(bgr), ptr(depth), ptr(self.dyna), ptr(self.label), ptr(self.mask), ptr(self.kps), self.cap,
                                      ptr(self.nkp), ptr(self.desc)), "sind_pipe_process")

    def process_dev(self, bgr_dev_ptr: int, depth_dev_ptr: int):
        """inputs already resident in HBM (e.g. torch tensors' data_ptr()); same layouts"""
        check(lib().sind_pipe_process_dev(self._h, ptr(bgr_dev_ptr), ptr(depth_dev_ptr), ptr(self.dyna), ptr(self.label), ptr(self.mask),
                                          ptr(self.kps), self.cap, ptr(self.nkp), ptr(self.desc)), "sind_pipe_process_dev")

    def submit_dev(self, bgr_dev_ptr: int, depth_dev_ptr: int) -> bool:
        """pipelined: enqueue a step; self.dyna/... receive the PREVIOUS step's results (returns False on the first call)"""
        have = C.c_int(0)
        check(lib().sind_pipe_submit_dev(self._h, ptr(bgr_dev_ptr), ptr(depth_dev_ptr), ptr(self.dyna), ptr(self.label), ptr(self.mask),
                                         ptr(self.kps), self.cap, ptr(self.nkp), ptr(self.desc), C.byref(have)), "sind_pipe_submit_dev")
        return bool(have.value)

    def flush(self) -> bool:
        have = C.c_int(0)
        check(lib().sind_pipe_flush(self._h, ptr(self.dyna), ptr(self.label), ptr(self.mask), ptr(self.kps), self.cap, ptr(self.nkp), ptr(self.desc),
                                    C.byref(have)), "sind_pipe_flush")
        return bool(have.value)

    def set_depth_ahead(self, on: bool):
        """schedule of the synchronous step: depth half of the tails underneath the dense flow (same results)"""
        check(lib().sind_pipe_set_depth_ahead(self._h, int(bool(on))), "sind_pipe_set_depth_ahead")

    def set_grow_share(self, quarters: int):
        """PEAC region grow: `quarters` of every four frames on the GPU, the rest on the host (same results); -1 = adaptive (default)"""
        check(lib().sind_pipe_set_grow_share(self._h, int(quarters)), "sind_pipe_set_grow_share")

    def grow_share(self) -> int:
        q = C.c_int(); check(lib().sind_pipe_get_grow_share(self._h, C.byref(q))); return q.value

    def set_kmeans_groups(self, groups: int):
        """batched k-means rounds as `groups` independent chains over groups of streams (1 .. min(4, streams // 8)); -1 = adaptive (default); same results"""
        check(lib().sind_pipe_set_kmeans_groups(self._h, int(groups)), "sind_pipe_set_kmeans_groups")

    def kmeans_groups(self) -> int:
        """groups of streams whose batched k-means rounds run as independent chains right now (0: no batched k-means)"""
        q = C.c_int(); check(lib().sind_pipe_get_kmeans_groups(self._h, C.byref(q)), "sind_pipe_get_kmeans_groups"); return q.value

    def host_info(self) -> dict:
        a = np.zeros(6, np.int32); check(lib().sind_pipe_host_info(self._h, ptr(a)), "sind_pipe_host_info")
        return dict(cpu_share=int(a[0]), workers=int(a[1]), cpu_tokens_max=int(a[2]), cores_usable=int(a[3]), cgroup_quota_cores=(int(a[4]) if a[4] >= 0 else None), local_world=int(a[5]))

    def get_state_bytes(self) -> int:
        return int(lib().sind_pipe_state_bytes(self._h))

    @staticmethod
    def state_bytes_for(width: int, height: int) -> int:
        """size of the state blob (DynaTail::state_bytes): four byte images + two 256-entry count tables + one flag"""
        return 4 * width * height + 2 * 256 * 4 + 4

    def get_state(self, stream: int = 0) -> np.ndarray:
        """inter-frame state of one stream (imgDynaLast / imgLabelLast / imgMaskHighErrorLast / k-means warm labels) as a u8 blob"""
        n = lib().sind_pipe_state_bytes(self._h); buf = np.empty(n, np.uint8)
        check(lib().sind_pipe_get_state(self._h, stream, ptr(buf), C.c_size_t(n)), "sind_pipe_get_state")
        return buf

    def set_state(self, stream: int, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, np.uint8)
        check(lib().sind_pipe_set_state(self._h, stream, ptr(blob), C.c_size_t(blob.size)), "sind_pipe_set_state")

    def set_state_hashing(self, on: bool = True):
        """every tail leaves a 128-bit fingerprint of its rolled inter-frame state per frame (read with state_hashes())"""
        check(lib().sind_pipe_set_state_hashing(self._h, int(bool(on))), "sind_pipe_set_state_hashing")

    def state_hashes(self) -> np.ndarray:
        """fingerprints of the step whose results were returned last: u64 [S, T, 2] ((0, 0) = the frame's tail did not run)"""
        out = np.zeros((self.S, self.T, 2), np.uint64)
        check(lib().sind_pipe_get_state_hashes(self._h, ptr(out), C.c_size_t(out.size)), "sind_pipe_get_state_hashes")
        return out

    def set_active_frames(self, frames_per_stream):
        """next step only: the stateful tail of stream s runs for its first frames_per_stream[s] frames (None = all)"""
        if frames_per_stream is None:
            check(lib().sind_pipe_set_active_frames(self._h, None), "sind_pipe_set_active_frames"); return
        a = np.ascontiguousarray(frames_per_stream, np.int32); assert a.shape == (self.S,)
        check(lib().sind_pipe_set_active_frames(self._h, ptr(a)), "sind_pipe_set_active_frames")

    def reserve_retained(self, steps: int):
        """set buffers aside for `steps` retained steps (retain_next / replay)"""
        check(lib().sind_pipe_reserve_retained(self._h, int(steps)), "sind_pipe_reserve_retained")

    def retain_next(self, tag: int):
        """keep the phase-A outputs of the next submitted step under `tag` (for replay)"""
        check(lib().sind_pipe_retain_next(self._h, int(tag)), "sind_pipe_retain_next")

    def replay(self, tag: int, first, last):
        """the stateful tails of frames [first[s], last[s]) of the retained step `tag` again, from the streams' current states; results in self.dyna / ... (step layout)"""
        a = np.ascontiguousarray(first, np.int32); b = np.ascontiguousarray(last, np.int32); assert a.shape == (self.S,) and b.shape == (self.S,)
        check(lib().sind_pipe_replay(self._h, int(tag), ptr(a), ptr(b), ptr(self.dyna), ptr(self.label), ptr(self.mask), ptr(self.kps), self.cap, ptr(self.nkp), ptr(self.desc)),
              "sind_pipe_replay")

    def release_retained(self, tag: int = -1):
        check(lib().sind_pipe_release_retained(self._h, int(tag)), "sind_pipe_release_retained")

    def set_chain_max_streams(self, n: int):
        """ragged / replayed steps with at most n active streams run as per-stream chains instead of batched rounds (default 12, 0 = never)"""
        check(lib().sind_pipe_set_chain_max_streams(self._h, int(n)), "sind_pipe_set_chain_max_streams")

    def keypoints(self, s: int, t: int):
        k = s * self.T + t; n = self.nkp[k]; return self.kps[k, :n], self.desc[k, :n]

    def stats(self):
        st = np.zeros(6); nl = C.c_longlong(); ms = C.c_double(); by = C.c_double()
        check(lib().sind_pipe_stats(self._h, ptr(st), C.byref(nl), C.byref(ms), C.byref(by)))
        un = C.c_double(); sl = C.c_int(); sm = C.c_double()
        check(lib().sind_pipe_sor_stats(self._h, C.byref(nl), C.byref(sm), C.byref(un), C.byref(by), C.byref(sl)))
        tw = C.c_double(); check(lib().sind_pipe_tail_wait_ms(self._h, C.byref(tw)))
        ol = C.c_longlong(); om = C.c_double(); ob = C.c_double(); check(lib().sind_pipe_sor_other_stats(self._h, C.byref(ol), C.byref(om), C.byref(ob)))
        return dict(front_ms=st[0], flow_ms=st[1], orb_ms=st[2], upload_ms=st[3], tails_ms=st[4], total_ms=st[5], tail_wait_ms=tw.value, sor_launches=nl.value, sor_ms=sm.value, sor_alg_bytes=by.value,
                    sor_union_ms=un.value, sor_slices=sl.value, sor_other_launches=ol.value, sor_other_ms=om.value, sor_other_alg_bytes=ob.value)


class PipelineGroup:
    """P independent pipelines on one GPU, each with a contiguous share of the S streams, driven concurrently (one host thread each; ctypes calls release the
    GIL) behind the interface of ONE Pipeline(S, T).  Why: a small step (a rank of a multi-GPU sequence job sees ~30 frame pairs per step) is a chain of
    ~2000 dependent launches whose latency, not the GPU's throughput, sets its time; two or three such chains side by side fill the gaps of each other.  The streams are
    independent (one reference DynaDetect instance each), so the results do not depend on the partition (tests/test_pipeline_gpu.py)."""

    def __init__(self, parts: int, streams: int, frames_per_step: int, *args, **kw):
        from concurrent.futures import ThreadPoolExecutor
        parts = max(1, min(parts, streams))
        self.S, self.T = streams, frames_per_step
        probe_w = kw.get("width", args[0] if len(args) > 0 else 640); probe_h = kw.get("height", args[1] if len(args) > 1 else 480)
        self.w, self.h = probe_w, probe_h
        nfeatures = kw.get("nfeatures", args[7] if len(args) > 7 else 1500)
        self.cap = 2 * nfeatures + 256
        B = streams * frames_per_step; shape = (streams, frames_per_step, probe_h, probe_w)
        self.dyna_pinned = None
        try:
            import torch
            if torch.cuda.is_available():
                self.dyna_pinned = torch.zeros(shape, dtype=torch.uint8).pin_memory()
        except ImportError:
            pass
        self.dyna = self.dyna_pinned.numpy() if self.dyna_pinned is not None else np.zeros(shape, np.uint8)
        self.label = np.zeros(shape, np.uint8); self.mask = np.zeros(shape, np.uint8)
        self.kps = np.zeros((B, self.cap), KP_DTYPE); self.nkp = np.zeros(B, np.int32); self.desc = np.zeros((B, self.cap, 32), np.uint8)
        self.first = [streams * i // parts for i in range(parts + 1)]
        self.pipes = []
        for i in range(parts):
            s0, s1 = self.first[i], self.first[i + 1]; T = frames_per_step
            out = (self.dyna[s0:s1], self.label[s0:s1], self.mask[s0:s1], self.kps[s0 * T:s1 * T], self.nkp[s0 * T:s1 * T], self.desc[s0 * T:s1 * T])
            self.pipes.append(Pipeline(s1 - s0, frames_per_step, *args, _out=out, **kw))
        share = self.pipes[0].host_info()["cpu_share"]
        for p in self.pipes:
            check(lib().sind_pipe_set_cpu_share(p._h, max(2, -(-share // parts))), "sind_pipe_set_cpu_share")
        self._ex = ThreadPoolExecutor(parts) if parts > 1 else None

    def _all(self, fn):
        if self._ex is None:
            return [fn(0, self.pipes[0])]
        return list(self._ex.map(lambda ip: fn(*ip), enumerate(self.pipes)))

    def _of(self, s):
        for i, p in enumerate(self.pipes):
            if self.first[i] <= s < self.first[i + 1]:
                return p, s - self.first[i]
        raise IndexError(s)

    def _ptrs(self, i, b, d):
        off = self.first[i] * self.T * self.h * self.w
        return b + off * 3, d + off * 2

    def close(self):
        for p in getattr(self, "pipes", []):
            p.close()
        if getattr(self, "_ex", None):
            self._ex.shutdown(); self._ex = None

    __del__ = close

    def prime(self, s, a, b): p, k = self._of(s); p.prime(k, a, b)
    def process_dev(self, b, d): self._all(lambda i, p: p.process_dev(*self._ptrs(i, b, d)))
    def process(self, bgr, depth): self._all(lambda i, p: p.process(bgr[self.first[i]:self.first[i + 1]], depth[self.first[i]:self.first[i + 1]]))
    def submit_dev(self, b, d): return all(self._all(lambda i, p: p.submit_dev(*self._ptrs(i, b, d))))
    def flush(self): return all(self._all(lambda i, p: p.flush()))
    def set_state_hashing(self, on=True): [p.set_state_hashing(on) for p in self.pipes]
    def set_depth_ahead(self, on): [p.set_depth_ahead(on) for p in self.pipes]
    def set_chain_max_streams(self, n): [p.set_chain_max_streams(n) for p in self.pipes]
    def state_hashes(self): return np.concatenate([p.state_hashes() for p in self.pipes])
    def get_state_bytes(self): return self.pipes[0].get_state_bytes()
    def get_state(self, s=0): p, k = self._of(s); return p.get_state(k)
    def set_state(self, s, blob): p, k = self._of(s); p.set_state(k, blob)
    def reserve_retained(self, n): [p.reserve_retained(n) for p in self.pipes]
    def retain_next(self, tag): [p.retain_next(tag) for p in self.pipes]
    def release_retained(self, tag=-1): [p.release_retained(tag) for p in self.pipes]
    def keypoints(self, s, t): k = s * self.T + t; n = self.nkp[k]; return self.kps[k, :n], self.desc[k, :n]
    def grow_share(self): return self.pipes[0].grow_share()
    def kmeans_groups(self): return self.pipes[0].kmeans_groups()
    def host_info(self): h = self.pipes[0].host_info(); h["pipelines"] = len(self.pipes); return h

    def set_active_frames(self, a):
        for i, p in enumerate(self.pipes):
            p.set_active_frames(None if a is None else np.asarray(a, np.int32)[self.first[i]:self.first[i + 1]])

    def replay(self, tag, first, last):
        f = np.asarray(first, np.int32); l = np.asarray(last, np.int32)
        self._all(lambda i, p: p.replay(tag, f[self.first[i]:self.first[i + 1]], l[self.first[i]:self.first[i + 1]]) if (l[self.first[i]:self.first[i + 1]] > f[self.first[i]:self.first[i + 1]]).any() else None)

    def stats(self):
        st = [p.stats() for p in self.pipes]; out = {}
        for k in st[0]:
            out[k] = max(x[k] for x in st) if k.endswith("_ms") and not k.startswith("sor") else sum(x[k] for x in st)
        return out
