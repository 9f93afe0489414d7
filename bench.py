#!/usr/bin/env python3
"""bench.py -- DynaDetect+ORB frame-pairs/s at 640x480 on MI355X (BASELINE.json metric).

One step = one pass of the batched pipeline over S streams x T frames (S*T frame pairs) of the synthetic TUM-shaped
RGB-D stream (TUM3 intrinsics, depth factor 5000, FAST 15/5, 1500 features; BASELINE.json configs[1] shape -- the real
TUM frames are not available offline).  Inputs are resident in HBM before the timed region (at most 12 distinct steps of input are
generated; a longer run cycles through them, the jump at the wrap is just another large-motion frame pair).  With --gpus N every rank
runs its own S streams (frames shard by stream, no data-path collective inside the hot path) and the per-frame dynamic
masks are gathered with one RCCL all_gather per step (north_star); value = all ranks' frame pairs / max-over-ranks time.

Prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def make_inputs(S, T, nsteps, seed=12345):
    """Base synthetic sequence + cheap per-stream variants (flips / gain) so that streams differ."""
    from sindslam_amd.synth import SyntheticStream
    F = T * nsteps + 2
    base_b, base_d = SyntheticStream(seed=seed).frames(0, F)
    bgr = np.empty((S, F) + base_b.shape[1:], np.uint8); depth = np.empty((S, F) + base_d.shape[1:], np.uint16)
    for s in range(S):
        b, d = base_b, base_d
        if s & 1: b, d = b[:, :, ::-1], d[:, :, ::-1]
        if s & 2: b, d = b[:, ::-1], d[:, ::-1]
        g = 1.0 - 0.04 * ((s >> 2) % 4)
        bgr[s] = np.clip(b.astype(np.float32) * g, 0, 255).astype(np.uint8) if g != 1.0 else b
        depth[s] = d
    return bgr, depth


def cpu_baseline(n_par, bgr, depth, gpu_dyna, gpu_kps, threads=8, n_timed=10):
    """Oracle ('port') DynaDetect + dilate + ORB on the host over a bounded sample of the same workload: the first `n_timed` frame
    pairs of the bench's first `threads` streams, one scalar oracle instance per thread (the reference pins its own loops to
    omp_set_num_threads(8), DynaDetect.cc:268) -> also the parity figures of the metric ("mask IoU vs CPU ref", ORB keypoints
    bit-exact) on the first n_par frames of each of those streams, which the GPU pipeline processed in its first step."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from sindslam_amd.synth import TUM3
    threads = max(1, min(threads, bgr.shape[0], os.cpu_count() or 1)); n_timed = min(max(n_par, n_timed), bgr.shape[1] - 2)
    O.lib()                                  # load before the clock starts; ctypes calls release the GIL
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda s: O.baseline_run(bgr[s, :n_timed + 2], depth[s, :n_timed + 2], TUM3, want_outputs=True), range(threads)))
    wall = time.perf_counter() - t0
    ious = []; kp_equal = 0; st = np.zeros(3); cpu_s = 0.0
    for s, (t, stage, dyna, kps) in enumerate(res):
        st += stage; cpu_s += t
        for i in range(n_par):
            a, r = gpu_dyna[s][i] == 255, dyna[i] == 255; u = np.logical_or(a, r).sum()
            ious.append(1.0 if u == 0 else float(np.logical_and(a, r).sum() / u))
            kp_equal += int(gpu_kps[s][i].tobytes() == kps[i].tobytes())
    base = {"value": threads * n_timed / wall, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{threads} streams x {n_timed} frame pairs of the synthetic 640x480 workload, one oracle instance per thread: {wall:.1f} s wall, "
                      f"{cpu_s:.1f} core-seconds (flow {st[0]:.1f}, tail {st[1]:.1f}, orb {st[2]:.1f}) = {threads * n_timed / cpu_s:.2f} pairs/s per core"}
    parity = {"mask_iou_mean": float(np.mean(ious)), "mask_iou_min": float(np.min(ious)), "orb_keypoints_bit_exact_frames": kp_equal, "frames": len(ious),
              "reference": "CPU oracle (parity unpinned: the reference ships no golden vectors and cannot be built here)"}
    return base, parity


def thread_cpu_seconds():
    """CPU seconds (user + system) of this process's live threads, summed by thread name (/proc/self/task/*/stat)."""
    tick = os.sysconf("SC_CLK_TCK"); acc = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        name = st[st.index("(") + 1: st.rindex(")")]; f = st[st.rindex(")") + 2:].split()
        acc[name] = acc.get(name, 0.0) + (int(f[11]) + int(f[12])) / tick
    return acc


def cgroup_throttle():
    """(periods, throttled periods, throttled microseconds) of this container's CPU quota (cgroup v2 cpu.stat), or None"""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return int(kv["nr_periods"]), int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(pairs_per_step):
    """HBM bytes per k_sor_fused launch from the committed rocprofv3 PMC passes (profiles/r01/v13_pmc_k_sor_fused.json: separate
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction), scaled to this batch; None if absent."""
    f = os.path.join(ROOT, "profiles", "r01", "v13_pmc_k_sor_fused.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    return d["hbm_bytes_per_launch_per_pair"] * pairs_per_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--frames-per-step", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=8, help="host threads of the cpu_baseline leg (one oracle instance and one stream each)")
    ap.add_argument("--host-input", action="store_true", help="hand over HOST buffers each step (sind_pipe_process, PCIe-inclusive rate; DESIGN.md 6) instead of HBM-resident inputs")
    ap.add_argument("--thread-cpu", action="store_true", help="print the CPU seconds the live threads used inside the timed region, by thread name (stderr)")
    ap.add_argument("--pipelined", action="store_true", help="software-pipeline consecutive steps (submit/flush); off by default: measured slower on MI355X")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker pool size (0 = library default, the GPU box's CPU share)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for CPU-side rehearsal)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    local = local % torch.cuda.device_count()          # rehearsal on a 1-GPU box: ranks share the card
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)     # backend "nccl" is RCCL on ROCm

    from sindslam_amd.pipeline import Pipeline
    from sindslam_amd.synth import TUM3
    S, T, K, Wm = args.streams, args.frames_per_step, args.steps, args.warmup
    nsteps = K + Wm
    ndata = min(nsteps, 12)                 # distinct steps of input kept in host + device memory (393 MB each); longer runs cycle through them
    bgr, depth = make_inputs(S, T, ndata, seed=12345 + rank)
    pipe = Pipeline(S, T, 640, 480, TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"], 1500, 1.2, 8, TUM3["ini_th"], TUM3["min_th"],
                    orb_gray_rgb_order=1, device=local, host_threads=args.host_threads)
    for s in range(S):
        pipe.prime(s, bgr[s, 1], bgr[s, 0])
    # inputs resident in HBM before the timed region, laid out [step][S][T]...
    dev_b = [torch.from_numpy(np.ascontiguousarray(bgr[:, 2 + i * T: 2 + (i + 1) * T])).cuda() for i in range(ndata)]
    dev_d = [torch.from_numpy(np.ascontiguousarray(depth[:, 2 + i * T: 2 + (i + 1) * T]).view(np.int16)).cuda() for i in range(ndata)]
    torch.cuda.synchronize()
    from sindslam_amd.parallel import gather_masks

    gbuf = {}

    def gather():
        if world > 1:   # RCCL gather of the per-frame dynamic masks over xGMI (page-locked source, persistent device buffers)
            m = pipe.dyna_pinned if pipe.dyna_pinned is not None else torch.from_numpy(pipe.dyna)
            if args.backend == "nccl":
                if "dev" not in gbuf:
                    gbuf["dev"] = torch.empty_like(m, device="cuda"); gbuf["out"] = torch.empty((world,) + tuple(m.shape), dtype=m.dtype, device="cuda")
                gbuf["dev"].copy_(m, non_blocking=True)
                gather_masks(gbuf["dev"], out=gbuf["out"])
            else:
                gather_masks(m)

    NPS = min(S, args.cpu_threads) if (world == 1 and not args.no_cpu_baseline) else 0

    def parity_sample():
        return [pipe.dyna[s].copy() for s in range(NPS)], [[pipe.keypoints(s, t)[0].copy() for t in range(T)] for s in range(NPS)]

    host_b = host_d = None
    if args.host_input:
        host_b = [np.ascontiguousarray(bgr[:, 2 + i * T: 2 + (i + 1) * T]) for i in range(ndata)]
        host_d = [np.ascontiguousarray(depth[:, 2 + i * T: 2 + (i + 1) * T]) for i in range(ndata)]
    first_dyna = first_kps = None
    for i in range(Wm):                     # warm-up: synchronous steps
        pipe.process_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()); gather()
        if i == 0:                          # the first streams' first T results, for the parity figures below
            first_dyna, first_kps = parity_sample()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    thr0 = cgroup_throttle()
    t0 = time.perf_counter(); c0 = time.process_time(); th0 = thread_cpu_seconds() if args.thread_cpu else None
    sor_ms = sor_bytes = sor_union = 0.0; sor_launches = 0; sor_slices = 1; stages = np.zeros(6)
    for i in range(Wm, Wm + K):             # timed: software-pipelined steps (phase A of step i overlaps the tails of step i-1)
        if args.host_input:
            pipe.process(host_b[i % ndata], host_d[i % ndata]); gather()
        elif not args.pipelined:
            pipe.process_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()); gather()
        elif pipe.submit_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()):
            gather()
        if first_dyna is None and not args.pipelined:     # --warmup 0: take the parity sample from the first timed step (a few MB copied)
            first_dyna, first_kps = parity_sample()
        st = pipe.stats()
        sor_ms += st["sor_ms"]; sor_bytes += st["sor_alg_bytes"]; sor_launches += st["sor_launches"]; sor_union += st["sor_union_ms"]; sor_slices = st["sor_slices"]
        stages += np.array([st["front_ms"], st["flow_ms"], st["orb_ms"], st["tails_ms"], st["total_ms"], st["upload_ms"]])
    if args.pipelined and pipe.flush():      # drain the last step inside the timed region
        gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    th1 = thread_cpu_seconds() if args.thread_cpu else None
    thr1 = cgroup_throttle()
    dt = time.perf_counter() - t0; cpu_busy = (time.process_time() - c0) / dt      # host cores this rank kept busy (all threads)
    if world > 1:
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    pairs = S * T * K * world
    if args.thread_cpu and rank == 0:       # short-lived threads (flow slices, ORB, octree) that ended before the second sample are not listed
        for name in sorted(th1, key=lambda n: -(th1[n] - th0.get(n, 0.0))):
            print(f"[thread-cpu] {name:16s} {(th1[name] - th0.get(name, 0.0)) / K * 1e3:9.1f} core-ms per step", file=sys.stderr)
        live = sum(th1[n] - th0.get(n, 0.0) for n in th1)
        print(f"[thread-cpu] {'(exited threads)':16s} {(cpu_busy * dt - live) / K * 1e3:9.1f} core-ms per step   total {cpu_busy * dt / K * 1e3:.1f}", file=sys.stderr)
    if rank == 0:
        # The batch runs as `sor_slices` slices on concurrent HIP streams, so solver launches overlap on the GPU.  achieved = algorithmic
        # bytes of all launches / time with at least one solver launch in flight (union of the HIP-event intervals of all slices on a
        # common time base); with one slice this is exactly bytes per launch / average launch duration.  The per-launch figures
        # (a launch that shares the GPU with the other slices) are reported next to it.
        achieved = sor_bytes / (sor_union * 1e-3) / 1e9 if sor_union > 0 else 0.0      # GB/s
        per_launch = sor_bytes / (sor_ms * 1e-3) / 1e9 if sor_ms > 0 else 0.0
        out = {
            "metric": "DynaDetect+ORB frame-pairs/sec at 640x480; mask IoU vs CPU ref", "value": pairs / dt, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "TUM fr3/walking_xyz-shaped synthetic RGB-D stream, 640x480, TUM3 intrinsics, FAST 15/5, 1500 features",
                       "streams_per_gpu": S, "frames_per_step": T, "frame_pairs_per_step": S * T * world, "parallelism": f"stream-sharded x{world}", "pipelined": bool(args.pipelined),
                       "inputs": "host buffers, H2D inside the timed region" if args.host_input else "resident in HBM"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": pmc_traffic(S * T / max(sor_slices, 1)),
                         "kernel": "k_sor_fused", "launches": sor_launches, "avg_launch_us": (sor_ms * 1e3 / sor_launches) if sor_launches else None,
                         "alg_bytes_per_launch": (sor_bytes / sor_launches) if sor_launches else None,
                         "concurrent_launches": sor_slices, "achieved_per_launch": per_launch, "solver_busy_ms_per_step": sor_union / K},
            "stage_ms_per_step": {"front": stages[0] / K, "dense_flow": stages[1] / K, "orb_front": stages[2] / K, "tails": stages[3] / K, "host_upload": stages[5] / K, "total": stages[4] / K},
            "host_cores_busy": cpu_busy,
            "cpu_quota": None if not (thr0 and thr1) else {"periods": thr1[0] - thr0[0], "throttled_periods": thr1[1] - thr0[1], "throttled_ms": (thr1[2] - thr0[2]) / 1e3},
        }
        if not args.no_cpu_baseline and world == 1 and first_dyna is not None:
            out["cpu_baseline"], out["parity"] = cpu_baseline(min(T, 4), bgr, depth, first_dyna, first_kps, threads=NPS)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
