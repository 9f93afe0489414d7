#!/usr/bin/env python3
"""bench.py -- DynaDetect+ORB frame-pairs/s at 640x480 on MI355X (BASELINE.json metric).

One step = one pass of the batched pipeline over S streams x T frames (S*T frame pairs).  Inputs are resident in HBM before the timed
region.  Two workloads:

  streams   (default at --gpus 1)  S independent TUM-shaped camera streams per GPU (BASELINE.json configs[1] shape: 640x480, TUM3
            intrinsics, depth factor 5000, FAST 15/5, 1500 features; the real TUM frames are not available offline); W untimed warm-up
            steps, K timed steps, weak scaling.  On one GPU the line also carries `sequence`: the fixed-length job below, run after
            the timed region, i.e. the N = 1 point of the curve that --gpus N > 1 reports.
  sequence  (default at --gpus N > 1; BASELINE.json configs[3]) ONE synthetic RGB-D sequence of --sequence-frames (4000) frames sharded by
            frame into N*S lock-step chunks (sindslam_amd.sequence.plan_lockstep): every chunk processes K*T frames, chunk 0 from the first
            frame on, chunk g > 0 starts 24 frames before its first owned frame to rebuild the tail state.  The W warm-up steps are
            untimed and their state is dropped (all streams are primed again); the K timed steps are the WHOLE job, chunk warm-up
            frames included, software-pipelined, and after every step the per-frame dynamic masks of all ranks are gathered with one
            RCCL all_gather over xGMI into the sequence-ordered mask array every rank holds.  value = sequence frames / wall time:
            strong scaling (the sequence length does not grow with N).  Rank 0 then re-runs the first frames in the in-order
            ("exact") mode and reports the chunk-seam IoU and the exact mode's own rate.

--gpus N without a launcher (WORLD_SIZE unset): this process starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set)
BEFORE anything touches torch or the GPU and only waits for them; under torch.distributed.run the ranks are already there.
Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TUM_SEQUENCE_FRAMES = 830       # frames of the one-GPU `sequence_tum_length` leg: TUM fr3/walking_xyz has ~820-860 associated frames
SEQ_BASE_FRAMES = 50            # generated frames of the sequence workload, walked forth and back
SEQ_WARMUP_FRAMES = 16          # state warm-up frames of a chunk: most rebuilt states equal the sequential one after ~16 frames; the others are found by the seam verification and repaired

# BASELINE.json configs that fit one GPU; `tum3` is the one the metric is quoted on
CONFIGS = {
    "tum3": dict(width=640, height=480, intr="TUM3", rgb=1, flow_max_levels=0, streams=128,
                 workload="TUM fr3/walking_xyz-shaped synthetic RGB-D stream, 640x480, TUM3 intrinsics, FAST 15/5, 1500 features"),
    "bonn": dict(width=640, height=480, intr="BONN", rgb=1, flow_max_levels=0, streams=128,
                 workload="Bonn rgbd-dynamic-shaped synthetic RGB-D stream, 640x480, Bonn intrinsics, FAST 20/7, 1500 features"),
    "d455_720p": dict(width=1280, height=720, intr="D455", rgb=0, flow_max_levels=3, streams=48,
                      workload="D455-shaped synthetic RGB-D stream, 1280x720, D455 intrinsics x2, depth factor 1000, FAST 20/7, 1500 features, 3-level flow pyramid"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)            # a pipelined run ends with one drained tail phase (~160 ms): 12 steps keep it at 3 % of the timed region
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="tum3", help="BASELINE.json config (default: the one the metric is quoted on)")
    ap.add_argument("--flow-levels", type=int, default=None, help="experiment: cap the DeepFlow pyramid at this many levels (NOT the config's workload: the line says so in config.flow_pyramid_levels)")
    ap.add_argument("--workload", choices=["auto", "streams", "sequence"], default="auto", help="auto: streams at 1 GPU, sequence (frame-sharded, RCCL mask gather) at N > 1")
    ap.add_argument("--streams", type=int, default=0, help="streams (= sequence chunks) per GPU; 0 = the config's default")
    ap.add_argument("--frames-per-step", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=8, help="host threads of the first cpu_baseline setting (the reference pins its OpenMP loops to 8)")
    ap.add_argument("--sequence-frames", type=int, default=4000, help="sequence workload: length of the ONE sequence (BASELINE.json configs[3]: 4000 frames); fixed as --gpus grows = strong scaling")
    ap.add_argument("--seq-warmup-frames", type=int, default=SEQ_WARMUP_FRAMES, help="sequence workload: frames every chunk after the first starts early to rebuild the inter-frame state (speculation; the seams are verified and repaired)")
    ap.add_argument("--retain-frames", type=int, default=-1, help="sequence workload: the steps holding the first N owned frames of every chunk keep their phase-A outputs, so that a repair run "
                                                                  "re-runs only the stateful tails of those frames (sind_pipe_replay); -1 = every step (default: how long a runner needs "
                                                                  "depends on the data), 0 = repair runs re-process whole frames")
    ap.add_argument("--chain-max", type=int, default=-1, help="sequence workload: replayed steps with at most this many live runners run as per-stream chains (-1 = library default 12, 0 = always rounds)")
    ap.add_argument("--repair-streams", type=int, default=0, help="sequence workload: runners of the repair pipeline (0 = half the chunks of a GPU, 2..16)")
    ap.add_argument("--repair-frames-per-step", type=int, default=4)
    ap.add_argument("--seq-driver", choices=["cabi", "python"], default="cabi", help="sequence workload: cabi = the C++ driver behind sind_seq_* (csrc/host/seq.cpp; default), "
                                                                                       "python = sindslam_amd/sequence.py VerifiedChunks (the same algorithm over torch.distributed)")
    ap.add_argument("--no-verify", action="store_true", help="sequence workload: keep the speculative chunk results (round-3 behaviour; masks not identical behind some seams)")
    ap.add_argument("--exact-leg-frames", type=int, default=0, help="sequence workload: frames of the in-order re-run the chunked masks are compared with (0 = chunk 0, chunks 1-2 and what else fits into 320-480 frames)")
    ap.add_argument("--flow-slices", type=int, default=0, help="experiment: dense-flow slices of a step (sind_pipe_config.flow_slices; 0 = the library's rule by step size)")
    ap.add_argument("--flow-opts-off", type=int, default=0, help="experiment: bit 0 = no k_coarse_chain, bit 1 = no k_sor_tile, bit 2 = no k_level_up, bit 3 = k_sor_stream instead of k_sor_wave, bits 8.. = waves per launch k_sor_wave cuts its row bands for (sind_pipe_config.flow_opts_off; same results)")
    ap.add_argument("--pipelines", type=int, default=0, help="independent pipelines a step is cut into on one GPU (experiment; 0 = one; results do not depend on it)")
    ap.add_argument("--no-n1-leg", action="store_true", help="sequence workload on N > 1 ranks: skip the one-rank run of the same job on rank 0 after the timed region (sequence.n1_value)")
    ap.add_argument("--no-tum-leg", action="store_true", help="streams workload on one GPU: skip the TUM-length single sequence (line field `sequence_tum_length`)")
    ap.add_argument("--no-sequence-leg", action="store_true", help="streams workload on one GPU: skip the fixed-length sequence job that is run after the timed region (line field `sequence`)")
    ap.add_argument("--no-dropin-leg", action="store_true", help="streams workload on one GPU: skip the one-frame-at-a-time run through the two classes' C ABI (line field `dropin`)")
    ap.add_argument("--no-small-step-leg", action="store_true", help="streams workload on one GPU: skip the 32-pair steps (line field `small_step`)")
    ap.add_argument("--dropin-frames", type=int, default=120)
    ap.add_argument("--no-exact-leg", action="store_true", help="sequence workload: skip the in-order re-run of the first chunks (seam IoU, exact-mode rate)")
    ap.add_argument("--host-input", action="store_true", help="hand over HOST buffers each step (sind_pipe_process, PCIe-inclusive rate; DESIGN.md 6) instead of HBM-resident inputs")
    ap.add_argument("--thread-cpu", action="store_true", help="print the CPU seconds the live threads used inside the timed region, by thread name (stderr)")
    ap.add_argument("--sync", action="store_true", help="synchronous steps (phase A, then the tails) instead of the default software pipeline in which phase A of step i+1 "
                                                     "overlaps the tails of step i (sind_pipe_submit_dev / flush; all K steps are drained inside the timed region)")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker pool size (0 = library default, the GPU box's CPU share)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for CPU-side rehearsal)")
    ap.add_argument("--collective", choices=["torch", "cabi"], default="torch", help="mask gather through torch.distributed (default) or through the library's own C-ABI collective "
                                                                                    "(sind_pipe_gather_masks on RCCL; torch.distributed then only hands the 128-byte id to the ranks)")
    ap.add_argument("--collective-at-1", action="store_true", help="with one rank, still create the process group and run the per-step mask gather (RCCL calls at world size 1)")
    ap.add_argument("--rendezvous-only", action="store_true", help="ranks only meet, count themselves and exit (launcher test, needs no GPU)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args) -> int:
    """Parent of a --gpus N run without a launcher: start N rank processes, never touch torch / the GPU here, exit with the first failure."""
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT", 29400 + os.getpid() % 2000))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    print(f"[bench launcher] pid {os.getpid()} spawned {n} ranks {[p.pid for p in procs]}; torch imported in the launcher: {'torch' in sys.modules}", file=sys.stderr, flush=True)
    rc = 0; left = list(procs)
    while left:
        for p in list(left):
            r = p.poll()
            if r is None:
                continue
            left.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in left:           # a dead rank would leave the others in a collective forever: end exactly the processes started here
                    q.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------------------------ inputs
_GEN = {}


def _gen_frame(a):
    from sindslam_amd.synth import SyntheticStream
    import sindslam_amd.synth as SY
    w, h, seed, intr, t = a
    if a[:4] not in _GEN:                    # one generator (textures) per worker process
        _GEN[a[:4]] = SyntheticStream(w, h, seed, getattr(SY, intr))
    return _GEN[a[:4]].frame(t)


def base_frames(cfg, n, seed, nseeds=1):
    """n consecutive frames of the synthetic generator (for nseeds > 1: of nseeds different scenes, stacked on a leading axis), produced by a few
    forked workers (numpy only -- called before torch / HIP exist)"""
    import multiprocessing as mp
    import numpy as np
    if nseeds > 1:
        jobs = [(cfg["width"], cfg["height"], seed + 1000 * k, cfg["intr"], t) for k in range(nseeds) for t in range(n)]
        nw = max(1, min(8, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
        with mp.get_context("fork").Pool(nw) as pool:
            fr = pool.map(_gen_frame, jobs, chunksize=1)
        b = np.stack([f[0] for f in fr]); d = np.stack([f[1] for f in fr])
        return b.reshape((nseeds, n) + b.shape[1:]), d.reshape((nseeds, n) + d.shape[1:])
    jobs = [(cfg["width"], cfg["height"], seed, cfg["intr"], t) for t in range(n)]
    nw = max(1, min(8, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    if nw > 1:
        with mp.get_context("fork").Pool(nw) as pool:
            fr = pool.map(_gen_frame, jobs, chunksize=1)
    else:
        fr = [_gen_frame(j) for j in jobs]
    return np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])


def stream_variants(base_b, base_d, S):
    """per-stream inputs: stream s shows scene s % (number of scenes) (base_b / base_d may carry a leading scene axis), mirrored / dimmed by the
    higher bits of s, so that the streams differ (3 scenes x 4 flips x 4 gains = 48 distinct streams)"""
    import numpy as np
    scenes = base_b.shape[0] if base_b.ndim == 5 else 1
    if base_b.ndim == 4: base_b, base_d = base_b[None], base_d[None]
    F = base_b.shape[1]
    bgr = np.empty((S, F) + base_b.shape[2:], np.uint8); depth = np.empty((S, F) + base_d.shape[2:], np.uint16)
    for s0 in range(S):
        b, d = base_b[s0 % scenes], base_d[s0 % scenes]; s = s0 // scenes
        if s & 1: b, d = b[:, :, ::-1], d[:, :, ::-1]
        if s & 2: b, d = b[:, ::-1], d[:, ::-1]
        g = 1.0 - 0.04 * ((s >> 2) % 4)
        bgr[s0] = np.clip(b.astype(np.float32) * g, 0, 255).astype(np.uint8) if g != 1.0 else b
        depth[s0] = d
    return bgr, depth


def pingpong(f, P):
    """frame f of an arbitrarily long sequence made of P base frames walked forth and back (consecutive frames stay neighbours)"""
    m = f % (2 * P - 2)
    return m if m < P else 2 * P - 2 - m


# ------------------------------------------------------------------------------------------------------------------ CPU baseline
def host_cpu_info():
    """nproc, and the cgroup-v2 CPU quota of this container (cores), if any"""
    info = {"nproc": os.cpu_count(), "cgroup_cpu_max_cores": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            info["cgroup_cpu_max_cores"] = int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    return info


def cpu_baseline(frames_of_stream, intr, cfg, n_par, gpu_dyna, gpu_kps, threads=8, n_streams=8):
    """Oracle ('port') DynaDetect + dilate + ORB on the host over a bounded sample of the same workload: one scalar oracle
    instance and one of the bench's streams per thread.  Setting A: `threads` (8, the reference's omp_set_num_threads(8), DynaDetect.cc:268)
    threads x (5 warm-up + 7 timed) pairs = 56 timed pairs; setting B: all cores of the box's quota x (2 + 4) pairs.  Stage means follow the
    reference's own stdout timers (DynaDetect.cc:1421,1499,1518,1161,1644 + ORB).  Also the parity figures of the metric ("mask IoU vs
    CPU ref", ORB keypoints bit-exact) on the first n_par frames of each sampled stream, which the GPU pipeline processed in its first step."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.lib()                                  # load before the clock starts; ctypes calls release the GIL
    info = host_cpu_info()
    cores_all = int(info["cgroup_cpu_max_cores"] or info.get("sched_affinity") or info["nproc"] or 1)

    def run(nthreads, warm, timed, want):
        data = [frames_of_stream(s % n_streams, warm + timed + 2) for s in range(nthreads)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            res = list(ex.map(lambda s: O.baseline_run(data[s][0], data[s][1], intr, orb_gray_rgb_order=cfg["rgb"], want_outputs=want, warmup_pairs=warm,
                                                       flow_max_levels=cfg["flow_max_levels"]), range(nthreads)))
        wall = time.perf_counter() - t0
        timed_s = sum(r[0] for r in res); st = np.sum([r[1][3:9] for r in res], axis=0) / (nthreads * timed)
        rate = nthreads * timed / max(r[0] for r in res)            # pairs of all threads over the slowest thread's timed span
        return res, dict(value=rate, threads=nthreads, warmup_pairs_per_thread=warm, timed_pairs=nthreads * timed, wall_s=wall, pairs_per_core_second=nthreads * timed / timed_s,
                         stage_ms_per_pair={k: float(v * 1e3) for k, v in zip(O.STAGES, st)})

    threads = max(1, min(threads, cores_all))
    resA, a = run(threads, 5, 7, True)
    b = None
    if cores_all > threads:
        _, b = run(cores_all, 2, 4, False)
    ious = []; kp_equal = 0
    for s in range(min(threads, n_streams, len(gpu_dyna))):
        _, _, dyna, kps = resA[s]
        for i in range(n_par):
            g, r = gpu_dyna[s][i] == 255, dyna[i] == 255; u = np.logical_or(g, r).sum()
            ious.append(1.0 if u == 0 else float(np.logical_and(g, r).sum() / u))
            kp_equal += int(gpu_kps[s][i].tobytes() == kps[i].tobytes())
    base = {"value": a["value"], "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads x (5 warm-up + 7 timed) frame pairs of the bench's first {min(threads, n_streams)} streams, one scalar oracle instance per thread "
                      f"({a['wall_s']:.1f} s wall); second setting: all {cores_all} cores of the quota x (2 + 4) pairs" + ("" if b else " -- skipped, no more cores than threads"),
            "host": info, "threads_8": a, "all_cores": b}
    parity = {"sample_from": "first warm-up step (synchronous call); the pipelined timed steps produce the same bytes (tests/test_pipeline_gpu.py::test_pipelined_submit_equals_sync)",
              "mask_iou_mean": float(np.mean(ious)), "mask_iou_min": float(np.min(ious)), "orb_keypoints_bit_exact_frames": kp_equal, "frames": len(ious),
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


def pmc_profile():
    """The committed rocprofv3 PMC passes of the solver kernel for the default config (separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per
    the gfx950 correction): newest profiles/rNN/*pmc_k_sor*.json, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_k_sor*.json")), key=lambda f: (os.path.basename(os.path.dirname(f)), os.path.basename(f)))        # rNN, then vK_ (file times do not survive a checkout)
    if not files:
        return None
    d = json.load(open(files[-1])); d["file"] = os.path.relpath(files[-1], ROOT)
    return d


# ------------------------------------------------------------------------------------------------------------------ measurement helpers
class StepAcc:
    """sums of the per-step pipeline statistics (solver HIP-event brackets, stage times) over the timed steps"""
    def __init__(self):
        import numpy as np
        self.sor_ms = self.sor_bytes = self.sor_union = 0.0; self.sor_launches = 0; self.sor_slices = 1           # the solver of the large levels (k_sor_wave / k_sor_stream)
        self.other_ms = self.other_bytes = 0.0; self.other_launches = 0                                           # solver launches of the other kernels (tiles, one-workgroup levels)
        self.stages = np.zeros(6); self.tail_wait = 0.0; self.submit_wall = 0.0

    def add(self, st, pipelined):
        import numpy as np
        self.sor_ms += st["sor_ms"]; self.sor_bytes += st["sor_alg_bytes"]; self.sor_launches += st["sor_launches"]; self.sor_union += st["sor_union_ms"]; self.sor_slices = st["sor_slices"]
        self.other_ms += st.get("sor_other_ms", 0.0); self.other_bytes += st.get("sor_other_alg_bytes", 0.0); self.other_launches += st.get("sor_other_launches", 0)
        self.stages += np.array([st["front_ms"], st["flow_ms"], st["orb_ms"], st["tails_ms"], st["total_ms"], st["upload_ms"]])
        self.tail_wait += st["tail_wait_ms"] if pipelined else 0.0


def flow_level_pixels(width, height, max_levels=0):
    """pixels of every DeepFlow pyramid level of the flow grid (0.6 x the frame, then x 0.95 per level until a side is <= 25; deepflow.cpp)"""
    import numpy as np
    w, h = int(np.float32(0.6) * width), int(np.float32(0.6) * height); out = []
    while True:
        out.append(w * h)
        nw, nh = int(np.float32(w) * np.float32(0.95) + np.float32(0.5)), int(np.float32(h) * np.float32(0.95) + np.float32(0.5))
        if nw <= 25 or nh <= 25 or len(out) >= 200:
            break
        w, h = nw, nh
    return out[:max_levels] if max_levels else out


VALU_PEAK = 78.6e12      # float operations per second of the vector units: 1024 SIMDs x 32 lanes x 2.4 GHz, one wave64 v_fma / v_mul / v_add per 2 cycles (MI355X_MICROARCH.md,
                         # cycle constants).  Measured with every SIMD issuing (profiles/r05/valu_rate.txt): 63 T/s unpacked at the clock the chip holds, and a PACKED FP32 instruction
                         # (v_pk_mul / v_pk_add / v_pk_fma: two floats per lane) takes the SIMD twice as long as an unpacked one (31 - 34 T instructions/s = 62 - 68 T float operations/s):
                         # it is priced as TWO operations.  One constant for every VALU figure of the line.


def roofline_of(acc, K, dt, pairs_per_launch, config_name, cfg=None):
    """`roofline` object of the dominant kernel: the SOR solver of every pyramid level above 8192 pixels in slices of 80 pairs and more -- k_sor_wave (one-wave row
    pipelines, flow_wave.hip), or k_sor_stream when --flow-opts-off has bit 3.

    bound = "hbm" (SURVEY 8d).  achieved = REAL HBM bytes per launch (committed rocprofv3 PMC passes of this command: FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction)
    x the solver launches of the timed steps / the time with at least one of them in flight (HIP events on the launching streams, union over the concurrent slices);
    frac = achieved / 8 TB/s, <= 1 by construction.  The kernel runs five iterations per launch out of registers and LDS (temporal blocking), so it moves 40 B per pixel and
    launch where the 8d yardstick prices 5 x 44 B -- achieved_algorithmic / frac_algorithmic keep that yardstick and can exceed 1.  Where it stands against BOTH roofs is
    measured by two experiments on the pyramid alone (bound_experiments: the same loads and stores with a tenth of the arithmetic; the same arithmetic with every load hitting
    the cache); valu_frac_from_counts (SQ_INSTS_VALU per kept pixel update x updates / time / VALU_PEAK) and valu_busy (an occupancy counter) are reported beside it, never as `frac`."""
    busy_s = acc.sor_union * 1e-3
    alg = acc.sor_bytes / busy_s / 1e9 if busy_s > 0 else 0.0                                   # algorithmic GB/s, device level
    pmc = pmc_profile() if config_name == "tum3" else None
    kname = "k_sor_stream" if (cfg or {}).get("flow_opts_off", 0) & 8 else "k_sor_wave"
    bk = (pmc or {}).get("by_kernel", {}).get(kname)
    traffic = None
    if bk:
        traffic = (2.0 * bk["FETCH_SIZE_kb_per_launch"] + bk["WRITE_SIZE_kb_per_launch"]) * 1024.0 * pairs_per_launch / pmc["pairs_per_launch"]
    roof = {"kernel": kname + " (SOR solver of the pyramid levels above 8192 pixels, slices of 80 pairs and more; 5 iterations per launch)", "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
            "launches": acc.sor_launches, "avg_launch_us": (acc.sor_ms * 1e3 / acc.sor_launches) if acc.sor_launches else None,
            "concurrent_slices": acc.sor_slices, "solver_busy_ms_per_step": acc.sor_union / K,
            "traffic": traffic,
            "traffic_source": (pmc["file"] + " (committed rocprofv3 --pmc passes of this command, per pair x pairs per launch; not re-measured in this run)") if pmc else None,
            "achieved": None, "frac": None,
            "alg_bytes_per_launch": (acc.sor_bytes / acc.sor_launches) if acc.sor_launches else None,
            "achieved_algorithmic": alg, "frac_algorithmic": alg / 8000.0,
            "algorithmic_note": "SURVEY 8d yardstick: 44 B per pixel update; a launch of five register-resident iterations moves 40 B per pixel, so this figure can exceed the peak and is not `frac`",
            "other_solver_kernels": {"launches": acc.other_launches, "sum_ms_per_step": acc.other_ms / K, "alg_bytes_per_step": acc.other_bytes / K,
                                     "note": "k_sor_tile / k_sor_fused launches of the step (tiled levels of small slices, one-workgroup levels of 4 - 8 k pixels); the levels of at most 4096 pixels run inside k_coarse_chain"}}
    if traffic and acc.sor_launches and busy_s > 0:
        hbm = traffic * acc.sor_launches / busy_s                                                # real HBM bytes per second while the streaming solver is busy
        roof["achieved"] = hbm / 1e9; roof["frac"] = min(1.0, hbm / 8e12)
        roof["achieved_per_launch"] = traffic / (acc.sor_ms * 1e-3 / acc.sor_launches) / 1e9 if acc.sor_ms > 0 else None
        if cfg is not None:
            px = [n for n in flow_level_pixels(cfg["width"], cfg["height"], cfg["flow_max_levels"]) if n > 8192]
            comp = 40.0 * (sum(px) / len(px)) * pairs_per_launch                                # per streaming launch: every streamed level has the same number of launches
            roof["compulsory_bytes_per_launch"] = comp; roof["traffic_over_compulsory"] = traffic / comp
        roof["bound_note"] = ("frac = achieved / peak of HBM, the roof SURVEY 8d names, while the solver shares the GPU with the step's other kernels (k_coef_lanes alone moves 0.4 x the solver's bytes).  "
                              "The solver is a temporally blocked stencil (five iterations per launch out of registers and LDS) that moves traffic_over_compulsory x its compulsory 40 B per pixel and launch; "
                              "alone on the GPU it needs bound_experiments.avg_launch_us per launch of 512 pairs against a memory floor and a compute floor measured with the same kernel (DESIGN.md 3.1)")
        if kname == "k_sor_wave" and pmc.get("bound_experiments"):
            be = dict(pmc["bound_experiments"]); per_pair = pmc["hbm_bytes_per_launch_per_pair"]
            be["frac_alone"] = per_pair * 512 / (be["avg_launch_us"] * 1e-6) / 8e12           # real bytes of a launch of 512 pairs / its measured duration / peak
            be["frac_alone_note"] = "committed measurement of the pyramid alone (one slice of 512 pairs), not re-measured in this run"
            roof["bound_experiments"] = be
    else:
        roof["bound_note"] = "no committed PMC passes for this config or no streaming launches in the timed steps (slices below 80 pairs run the tiled kernels): see achieved_algorithmic"
    # VALU side, from COUNTS: pixel updates (algorithmic bytes / 44 B) x VALU instructions per kept update (SQ_INSTS_VALU x 64 lanes / updates; a packed instruction counted as two) / time / VALU_PEAK
    ipu = (pmc or {}).get("valu_ops_per_pixel_update"); ipu2 = (pmc or {}).get("valu_slots_per_pixel_update", ipu)
    roof["useful_ops_per_update"] = 32; roof["instr_per_update"] = ipu; roof["valu_peak_ops_per_s"] = VALU_PEAK
    if busy_s > 0 and ipu2:
        roof["valu_frac_from_counts"] = (acc.sor_bytes / 44.0) * ipu2 / VALU_PEAK / busy_s
        roof["valu_note"] = ("instr_per_update VALU instructions per kept pixel update (counter-measured; 32 float operations are the arithmetic of an update, no FMA by contract), priced at "
                             "%.1f issue slots per kept update (idle lanes of the 128-column strips and the halo columns / rows of every cut included; a packed instruction would count twice: "
                             "profiles/r05/valu_rate.txt) against VALU_PEAK" % ipu2)
    if pmc and pmc.get("valu_busy_measured") is not None:
        roof["valu_busy"] = pmc["valu_busy_measured"]; roof["valu_busy_source"] = pmc.get("sq_counters")
        roof["valu_busy_note"] = "occupancy counter (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES), not achieved / peak"
    return roof


def host_load(c0, t0, thr0):
    """host cores this rank kept busy since (c0, t0) and the cgroup quota's throttle counters since thr0, as four floats (all-gathered over the ranks)"""
    dt = time.perf_counter() - t0; thr1 = cgroup_throttle()
    q = [float(thr1[i] - thr0[i]) for i in range(3)] if (thr0 and thr1) else [-1.0, -1.0, -1.0]
    return [(time.process_time() - c0) / dt, q[0], q[1], q[2] / 1e3 if q[2] >= 0 else -1.0]


def gather_host_load(mine, pg, comm_dev):
    """[cores busy, quota periods, throttled periods, throttled ms] of every rank"""
    import torch
    import torch.distributed as dist
    if not pg:
        return [mine]
    t = torch.tensor(mine, dtype=torch.float64, device=comm_dev); parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [[float(x) for x in p.cpu()] for p in parts]


def host_info_list(pipe):
    """the pipeline's own sizing of its host side (sind_pipe_host_info) as floats, appended to a rank's host_load vector"""
    h = pipe.host_info()
    return [float(h["cpu_share"]), float(h["workers"]), float(h["cores_usable"]), float(-1 if h["cgroup_quota_cores"] is None else h["cgroup_quota_cores"]), float(h["local_world"])]


def host_load_fields(loads):
    quota = lambda l: None if l[1] < 0 else {"periods": int(l[1]), "throttled_periods": int(l[2]), "throttled_ms": l[3]}
    sizing = lambda l: None if len(l) < 9 else {"cpu_share": int(l[4]), "pool_workers": int(l[5]), "cores_usable": int(l[6]), "cgroup_quota_cores": None if l[7] < 0 else int(l[7]), "ranks_on_node": int(l[8]),
                                                "rule": "share = min(cores usable, cgroup quota) / ranks on the node, at most 16, at least 4 without a quota / 2 with one (the shares of all ranks stay inside the quota)"}
    return {"host_cores_busy": loads[0][0], "cpu_quota": quota(loads[0]),
            "host_by_rank": [{"rank": r, "host_cores_busy": l[0], "cpu_quota": quota(l), "sizing": sizing(l)} for r, l in enumerate(loads)]}


def pipelines_for(S, T, asked=0):
    """independent pipelines a step of S x T frame pairs is cut into (sindslam_amd.pipeline.PipelineGroup): small steps are chains of dependent launches whose
    latency sets the step time; two or three chains side by side were expected to fill each other's gaps, but measured they do not (profiles/r04/small_steps.txt:
    651 -> 561 -> 502 pairs/s at 32 pairs per step for 1 -> 2 -> 3 pipelines), so the default stays one pipeline"""
    if asked > 0:
        return max(1, min(asked, S))
    return 1          # measured (profiles/r04/small_steps.txt): at 30-96 pairs per step one pipeline is the fastest; the group exists for experiments (--pipelines N)


def make_pipeline(cfg, intr, S, T, local, host_threads=0, parts=1):
    from sindslam_amd.pipeline import Pipeline, PipelineGroup
    a = (S, T, cfg["width"], cfg["height"], intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], 1500, 1.2, 8, intr["ini_th"], intr["min_th"])
    kw = dict(orb_gray_rgb_order=cfg["rgb"], device=local, host_threads=host_threads, flow_max_levels=cfg["flow_max_levels"], flow_slices=cfg.get("flow_slices", 0), flow_opts_off=cfg.get("flow_opts_off", 0))
    return PipelineGroup(parts, *a, **kw) if parts > 1 else Pipeline(*a, **kw)


def sequence_streams(world, frames, cfg_streams, steps, warm=None):
    """chunks per GPU of the fixed-length sequence when --streams is not given.  Every chunk after the first re-processes `warm` state warm-up frames, so more chunks mean
    more frames per step; fewer chunks mean more frames PER CHUNK per step, and the stateful tails of a chunk's frames are a serial chain (k-means from the previous frame's
    merged labels).  Measured on one MI355X at ~30 pairs per step (profiles/r05/rank_step_shapes.txt: what a rank of an 8-GPU job sees with the driver's 20 steps): 4 chunks x 8
    frames 509 pairs/s, 5 x 6 543, 6 x 5 658, 8 x 4 765, 10 x 3 783, 16 x 2 874 -- the step is max(dense flow of S * T pairs, T tails in a row); with the rounds split
    into depth and flow halves (profiles/r05/ab_split_rounds.txt) 5 x 6 670, 8 x 4 820, 16 x 2 unchanged.  So: among the chunk counts
    around the one that keeps the warm-up at about a quarter of the sequence, the one whose lock-step plan has the shortest estimated step (flow ~ 20 + 0.55 ms per pair, a round
    ~ 7 ms per frame of a chunk; fitted to those tables and to the 512-pair headline), ties to the plan that processes fewer frames."""
    from sindslam_amd.sequence import plan_lockstep
    warm = SEQ_WARMUP_FRAMES if warm is None else warm
    s0 = 4
    while s0 * 2 <= cfg_streams and s0 * 2 * world * 4 * max(warm, 8) <= frames:
        s0 *= 2
    best = None
    for s in range(max(4, (3 * s0) // 4), min(max(2 * s0, 13), cfg_streams + 1)):
        plan = plan_lockstep(frames, world * s, steps, warm)
        est = max(20.0 + 0.55 * s * plan.T, 7.0 * plan.T + 5.0)
        key = (est, plan.processed_total)
        if best is None or key < best[0]:
            best = (key, s)
    return best[1]


def dropin_leg(cfg, intr, frames_b, frames_d, device, n_frames=120):
    """The reference's own call pattern (rgbd_tum_noros.cc:131-139, Frame.cc:308): ONE camera, one frame per call, HOST pointers, in order, through the C ABI the two class
    shims bind (sind_dyna_detect + sind_dyna_dilate15 + sind_orb_extract).  frames_b / frames_d: host frames of one stream, walked forth and back.  The ORB input gray is
    formed by the caller before the clock (Tracking::GrabImageRGBD does it on the host in the reference, src/Tracking.cc:246-259); DynaDetect's own three gray conversions,
    both uploads and all downloads are inside."""
    import numpy as np
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.orb import ORBextractor
    P = frames_b.shape[0]
    cb, cr = (4899, 1868) if cfg["rgb"] else (1868, 4899)
    gray = [((b[..., 0].astype(np.int32) * cb + b[..., 1].astype(np.int32) * 9617 + b[..., 2].astype(np.int32) * cr + 8192) >> 14).astype(np.uint8) for b in frames_b]
    dd = DynaDetect(np.ascontiguousarray(frames_b[1]), np.ascontiguousarray(frames_b[0]), intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], device=device, debug=False)
    if cfg["flow_max_levels"]:
        dd.set_flow_max_levels(cfg["flow_max_levels"])
    orb = ORBextractor(1500, 1.2, 8, intr["ini_th"], intr["min_th"], device=device)
    td, tm, to = [], [], []; warm = 10
    for f in range(2, 2 + warm + n_frames):
        k = pingpong(f, P); b = np.ascontiguousarray(frames_b[k]); d = np.ascontiguousarray(frames_d[k])
        t0 = time.perf_counter(); dyna, _ = dd.DetectDynaArea(b, d, f)
        t1 = time.perf_counter(); mask = dd.dilate15(dyna)
        t2 = time.perf_counter(); orb(gray[k], mask)
        t3 = time.perf_counter()
        if f == 1 + warm:
            dd.timing(True)
        if f >= 2 + warm:
            td.append(t1 - t0); tm.append(t2 - t1); to.append(t3 - t2)
    st = dd.timing(); dd.close(); orb.close()
    tot = np.array(td) + np.array(tm) + np.array(to)
    return {"frames": len(td), "ms_per_frame": float(tot.mean() * 1e3), "fps": float(1.0 / tot.mean()), "ms_per_frame_max": float(tot.max() * 1e3),
            "detect_ms": float(np.mean(td) * 1e3), "dilate15_ms": float(np.mean(tm) * 1e3), "orb_ms": float(np.mean(to) * 1e3), "detect_stages_ms": st,
            "first_measurement": {"ms_per_frame": 29.1, "fps": 34.4, "where": "profiles/r05/dropin_first_measurement.json (round-4 code, same loop)"},
            "note": "in-order sind_dyna_detect + sind_dyna_dilate15 + sind_orb_extract at B = 1 with host pointers; the depth half of a frame runs beside its dense flow "
                    "(own stream and host thread), the large-motion candidate flow rides along as a batch of two"}


def small_step_leg(cfg, intr, bgr, depth, local, host_threads, S2=16, T2=2, steps=12, warm=3):
    """what a rank of an 8-GPU sequence job sees with the driver's --steps 20: ~32 frame pairs per step.  Same pipelined loop as the headline, S2 streams x T2 frames."""
    import numpy as np
    import torch
    pipe = make_pipeline(cfg, intr, S2, T2, local, host_threads)
    for s_ in range(S2):
        pipe.prime(s_, bgr[s_, 1], bgr[s_, 0])
    nd = min(steps + warm, (bgr.shape[1] - 2) // T2)
    db = [torch.from_numpy(np.ascontiguousarray(bgr[:S2, 2 + i * T2: 2 + (i + 1) * T2])).cuda() for i in range(nd)]
    dd = [torch.from_numpy(np.ascontiguousarray(depth[:S2, 2 + i * T2: 2 + (i + 1) * T2]).view(np.int16)).cuda() for i in range(nd)]
    torch.cuda.synchronize()
    for i in range(warm):
        pipe.process_dev(db[i % nd].data_ptr(), dd[i % nd].data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter(); flow = tails = 0.0
    for i in range(warm, warm + steps):
        pipe.submit_dev(db[i % nd].data_ptr(), dd[i % nd].data_ptr()); st = pipe.stats(); flow += st["flow_ms"]; tails += st["tails_ms"]
    pipe.flush(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sl = pipe.stats()["sor_slices"]; pipe.close()
    return {"pairs": S2 * T2, "streams": S2, "frames_per_step": T2, "steps": steps, "value": S2 * T2 * steps / dt, "unit": "frame-pairs/s", "ms_per_step": dt / steps * 1e3,
            "dense_flow_ms_per_step": flow / steps, "tails_ms_per_step": tails / steps, "flow_slices": sl,
            "first_measurement": {"value": 679.3, "shape": "8 streams x 4 frames, one slice", "where": "profiles/r05/small_step_first_measurement.json (round-4 code)"}}


class BenchFrames:
    """frame source of the sequence workload (sindslam_amd.sequence): the generated base frames live in HBM, sequence position q (0 = first processed frame; -1, -2 =
    the two priming frames of chunk 0) shows generated frame pingpong(q + 2) -- an arbitrarily long sequence whose consecutive frames stay neighbours"""

    def __init__(self, base_b, base_d):
        import numpy as np
        import torch
        self.base_b = base_b; self.P = base_b.shape[0]
        self.bb = torch.from_numpy(base_b).cuda(); self.bd = torch.from_numpy(base_d.view(np.int16)).cuda()

    def fidx(self, q):
        return pingpong(int(q) + 2, self.P)

    def host_frame(self, q):
        return self.base_b[self.fidx(q)]

    def device_batch(self, pos):
        import torch
        idx = torch.tensor([[self.fidx(q) for q in row] for row in pos], device="cuda")
        b = self.bb[idx].contiguous(); d = self.bd[idx].contiguous(); torch.cuda.current_stream().synchronize()
        return b.data_ptr(), d.data_ptr(), (b, d)


def sequence_job(args, cfg, intr, base_b, base_d, rank, world, local, pg, comm_dev, S, K, Wm, exact_leg):
    """BASELINE.json configs[3]: ONE synthetic RGB-D sequence of --sequence-frames frames, sharded by frame into world * S lock-step chunks (chunk g = stream
    g % S of rank g // S; sindslam_amd.sequence.plan_lockstep).  Wm untimed steps warm the process up (their state is thrown away: every stream is primed
    again), then the timed region is the WHOLE job: the K lock-step steps -- the state warm-up frames of every chunk after the first included --
    software-pipelined, with one all_gather of the step's dynamic masks per step (RCCL over xGMI with --backend nccl) into the sequence-ordered mask array
    every rank holds, THEN the verification of every chunk seam by state fingerprints and the repair of the chunks whose rebuilt state is not the sequential
    one (sindslam_amd.sequence.VerifiedChunks; their corrected masks are exchanged once per round), so that the masks equal the sequential loop's.
    Returns (seconds [max over ranks], StepAcc, info dict, host load of every rank)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from sindslam_amd.parallel import gather_masks
    from sindslam_amd.sequence import VerifiedChunks, plan_lockstep
    H, W = cfg["height"], cfg["width"]
    SW = args.seq_warmup_frames
    plan = plan_lockstep(args.sequence_frames, world * S, K, SW); T = plan.T; n = world * S
    if S * T > 4096:
        raise SystemExit(f"sequence workload: {S} chunks x {T} frames per step is more than one step should hold; use more --steps, fewer --streams or a shorter --sequence-frames")
    src = BenchFrames(base_b, base_d); bb, bd = src.bb, src.bd; fidx = src.fidx
    parts = pipelines_for(S, T, args.pipelines)
    R = args.repair_streams or max(2, min(16, (S + 1) // 2)); Tr = max(1, args.repair_frames_per_step)
    verify = n > 1 and not args.no_verify
    # ONE rank: the C++ driver behind sind_seq_* owns both pipelines and runs plan, steps, verification and repairs (csrc/host/seq.cpp); this function only feeds it the
    # device-resident batches and reads the step's masks in the step hook.  Several ranks (under torch.distributed, which also carries the mask gather): the Python twin.
    use_cabi = args.seq_driver == "cabi" and parts == 1
    job = vc = rp = net = None
    if use_cabi:
        from sindslam_amd.seq import SeqJob, SeqNet
        if world > 1:
            # the driver's own exchange (32 bytes of fingerprints per chunk and round, a state blob per mismatching seam between ranks) runs over TCP between the ranks' hosts
            # -- torch.distributed is left with what the bench contract asks of it: the per-step mask gather, the barrier and the reductions of the timing
            net = SeqNet.tcp(rank, world, 20000 + (int(os.environ.get("MASTER_PORT", "29500")) * 7) % 20000, os.environ.get("MASTER_ADDR", "127.0.0.1") if os.environ.get("MASTER_ADDR", "127.0.0.1")[0].isdigit() else None)
        job = SeqJob(args.sequence_frames, S, W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["depth_factor"], 1500, 1.2, 8, intr["ini_th"], intr["min_th"],
                     orb_gray_rgb_order=cfg["rgb"], device=local, host_threads=args.host_threads, flow_max_levels=cfg["flow_max_levels"], steps=K, warmup=SW, repair_streams=R,
                     repair_frames_per_step=Tr, retain_frames=args.retain_frames if verify else 0, verify=verify, flow_slices=cfg.get("flow_slices", 0), flow_opts_off=cfg.get("flow_opts_off", 0), net=net)
        assert job.T == T and job.steps == K and job.chunks.tolist() == [[c.first, c.last, c.start] for c in plan.chunks], "the C++ plan is the Python plan"
        pipe = job.pipeline_view()
    else:
        pipe = make_pipeline(cfg, intr, S, T, local, args.host_threads, parts)
        # the runners that repair mismatching chunks: a second, small pipeline (created and warmed before the clock starts, like the main one)
        rp = make_pipeline(cfg, intr, R, Tr, local, args.host_threads) if verify else None
    if args.chain_max >= 0 and hasattr(pipe, "set_chain_max_streams"):
        pipe.set_chain_max_streams(args.chain_max)
    if not use_cabi:
        vc = VerifiedChunks(plan, S, pipe, rp, src, rank, world, retain_frames=args.retain_frames if rp is not None else 0)
    mine = plan.chunks[rank * S:(rank + 1) * S]

    def prime_all():
        if use_cabi:
            job.prime(); return
        vc.prime()
        if rp is not None:
            for j in range(R):
                rp.prime(j, src.host_frame(-1), src.host_frame(-2))
            rp.set_state_hashing(True)
    dev_b, dev_d = [], []
    for i in range(K):
        idx = torch.tensor([[fidx(c.start + i * T + t) for t in range(T)] for c in mine], device="cuda")
        dev_b.append(bb[idx].contiguous()); dev_d.append(bd[idx].contiguous())
    # dynamic masks of every processed frame of ALL chunks, on every rank: [chunk][step][t]
    seq_masks = torch.zeros((world, S, K, T, H, W), dtype=torch.uint8, device="cuda" if comm_dev == "cuda" else "cpu")
    gbuf = {}
    if use_cabi:
        start0 = mine[0].start; keep = {}
        step_pos = [np.array([[c.start + i * T + t for t in range(T)] for c in mine], np.int64).ravel() for i in range(K)]

        def batch_source(pos):              # the K lock-step batches were built before the clock; a runner's batch is gathered from the resident base frames
            i = (int(pos[0]) - start0) // T
            if len(pos) == S * T and 0 <= i < K and np.array_equal(pos, step_pos[i]):
                return dev_b[i].data_ptr(), dev_d[i].data_ptr()
            b_, d_, keep["last"] = src.device_batch(pos.reshape(-1, Tr) if len(pos) == R * Tr else pos.reshape(S, T))
            return b_, d_
        job.set_source(batch_source, lambda q: np.ascontiguousarray(src.host_frame(q)))
        # sink: only REPAIRED frames are copied out (nkp marks them); the step hook below takes a lock-step step's masks from the step's own page-locked array
        fix_dyna = np.zeros((args.sequence_frames + 1, H, W), np.uint8); fix_mark = np.full(args.sequence_frames + 1, -1, np.int32)
        job.set_outputs(args.sequence_frames + 1, dyna=fix_dyna, nkp=fix_mark); job.set_emit_main(False)
    prime_all()

    cabi = None
    if pg and args.collective == "cabi":     # the library's own collective (RCCL through the C ABI): torch.distributed only carries the 128-byte id to the ranks
        from sindslam_amd.parallel import Comm
        box = [Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        cabi = Comm(box[0], rank, world, local)

    def gather(step):
        m = torch.from_numpy(job.step_masks()) if use_cabi else (pipe.dyna_pinned if pipe.dyna_pinned is not None else torch.from_numpy(pipe.dyna))
        if cabi is not None:
            if "out" not in gbuf:
                gbuf["out"] = torch.empty((world,) + tuple(m.shape), dtype=m.dtype, device="cuda")
            out = cabi.gather_pipeline_masks(pipe, out=gbuf["out"])
            if comm_dev != "cuda":
                out = out.cpu()
        elif pg and comm_dev == "cuda":
            if "dev" not in gbuf:
                gbuf["dev"] = torch.empty_like(m, device="cuda"); gbuf["out"] = torch.empty((world,) + tuple(m.shape), dtype=m.dtype, device="cuda")
            gbuf["dev"].copy_(m, non_blocking=True)
            out = gather_masks(gbuf["dev"], out=gbuf["out"])
            torch.cuda.current_stream().synchronize()            # the upload is complete before the next step may rewrite the page-locked source
        elif pg:
            out = gather_masks(m)
        else:
            out = m[None]
        if step is not None:
            seq_masks[:, :, step].copy_(out)
            if cabi is not None and out.is_cuda:
                torch.cuda.current_stream().synchronize()        # the comm's own stream rewrites gbuf["out"] at the next step: the copy out of it must be done by then

    if use_cabi:
        job.warm(Wm)                        # untimed: the first steps of the job + one step of the repair pipeline (first-use allocations), results and state dropped
        if Wm or verify:
            prime_all()
    else:
        for i in range(Wm):                 # untimed: the first steps of the job, results and state dropped afterwards
            pipe.process_dev(dev_b[i % K].data_ptr(), dev_d[i % K].data_ptr()); gather(None)
        if rp is not None:                  # one untimed step of the repair pipeline as well (first-use allocations)
            inp = src.device_batch(np.tile(np.arange(Tr), (R, 1))); rp.process_dev(inp[0], inp[1]); del inp
        if Wm or rp is not None:
            prime_all()                     # priming resets a stream's tail state: the timed region starts the job from scratch
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    thr0 = cgroup_throttle(); t0 = time.perf_counter(); c0 = time.process_time()
    acc = StepAcc()

    step_wall = []

    def after_submit(i, seconds):
        acc.submit_wall += seconds; acc.add(pipe.stats(), True); step_wall.append(round(seconds * 1e3, 1))
    if use_cabi:
        job.set_hooks(on_step=gather, on_round=lambda r_: on_round_cabi())
        for i in range(K):
            ts_ = time.perf_counter(); job.submit(i); after_submit(i, time.perf_counter() - ts_)
        tf_ = time.perf_counter(); job.flush(); flush_ms = (time.perf_counter() - tf_) * 1e3
    else:
        vc.run_main(on_step=lambda i, p_: gather(i), inputs=lambda i: (dev_b[i].data_ptr(), dev_d[i].data_ptr()), after_submit=after_submit)
        flush_ms = vc.flush_seconds * 1e3
    t_main = time.perf_counter() - t0
    # ---- verify the chunk seams, repair the mismatching chunks (inside the clock); corrected masks replace the speculative ones on every rank
    fixes = []                              # (global chunk, position, mask) of this round

    def on_frame(s_, q, p_, j, t_):
        g = rank * S + s_; c = mine[s_]; i = q - c.start
        if world == 1:
            seq_masks[0, s_, i // T, i % T].copy_(torch.from_numpy(p_.dyna[j, t_]))
        else:
            fixes.append((g, i, p_.dyna[j, t_].copy()))

    def on_round():
        if world == 1:
            return
        cnt = torch.tensor([len(fixes)], dtype=torch.int64, device=comm_dev); cnts = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        m = max(int(c_.item()) for c_ in cnts)
        if m:
            ids = torch.full((m, 2), -1, dtype=torch.int64); blk = torch.zeros((m, H, W), dtype=torch.uint8)
            for k_, (g, i, mk) in enumerate(fixes):
                ids[k_, 0] = g; ids[k_, 1] = i; blk[k_] = torch.from_numpy(mk)
            ids = ids.to(comm_dev); blk = blk.to(comm_dev)
            all_ids = [torch.empty_like(ids) for _ in range(world)]; dist.all_gather(all_ids, ids)
            all_blk = gather_masks(blk)
            for r_ in range(world):
                for k_ in range(int(cnts[r_].item())):
                    g, i = int(all_ids[r_][k_, 0]), int(all_ids[r_][k_, 1])
                    seq_masks[g // S, g % S, i // T, i % T].copy_(all_blk[r_][k_])
        fixes.clear()
    def on_round_cabi():
        # the frames the runners of this round re-ran (the sink's marks): their masks replace the speculative ones -- on this rank directly, on the others through on_round's exchange
        for f in np.nonzero(fix_mark >= 0)[0]:
            q = int(f) - 1; g = next(g_ for g_, c in enumerate(plan.chunks) if c.first <= q < c.last); i = q - plan.chunks[g].start
            if world == 1:
                seq_masks[0, g, i // T, i % T].copy_(torch.from_numpy(fix_dyna[f]))
            else:
                fixes.append((g, i, fix_dyna[f].copy()))
            fix_mark[f] = -1
        on_round()
    if use_cabi:
        job.verify(); vstats = job.stats()
        rf_ = args.retain_frames if args.retain_frames > 0 else max(1, plan.processed - SW)
        n_retained = len(range(SW // T, min(K, (SW + rf_ - 1) // T + 1))) if (verify and args.retain_frames != 0) else 0
    else:
        vstats = vc.verify_and_repair(on_frame=on_frame, on_round=on_round) if rp is not None else dict(vc.stats); n_retained = len(vc.retained)
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    load = host_load(c0, t0, thr0) + host_info_list(pipe); dt = time.perf_counter() - t0
    if pg:
        tt = torch.tensor([dt], device=comm_dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    loads = gather_host_load(load, pg, comm_dev)
    grow_q = pipe.grow_share(); km_groups = pipe.kmeans_groups()
    owned = sum(c.last - c.first for c in plan.chunks)
    info = {"frames": plan.frames, "owned_frames": owned, "chunks": n, "chunks_per_gpu": S, "frames_per_step_per_chunk": T, "steps": K,
            "processed_frames_per_chunk": plan.processed, "state_warmup_frames": SW, "state_warmup_steps": -(-SW // T),
            "processed_frames": plan.processed_total, "warmup_overhead": plan.processed_total / owned - 1.0,
            "value": owned / dt, "value_excl_warmup": plan.processed_total / dt, "seconds": dt, "final_flush_ms": flush_ms, "region_grow_gpu_quarters": grow_q, "kmeans_groups": km_groups,
            "exact": verify or n == 1, "pipelines_per_gpu": parts, "host_cores_busy": loads[0][0], "submit_wall_ms_by_step": step_wall,
            "cpu_quota": (None if loads[0][1] < 0 else {"periods": int(loads[0][1]), "throttled_periods": int(loads[0][2]), "throttled_ms": loads[0][3]}),
            "stage_ms_per_step": {"dense_flow": acc.stages[1] / K, "tails": acc.stages[3] / K, "total": acc.stages[4] / K, "tails_wait_after_phase_a": acc.tail_wait / K},
            "verify": {"seams": vstats["seams"], "mismatched_seams": vstats["mismatched_seams"], "rounds": vstats["rounds"], "repaired_chunks": vstats["repaired_chunks"],
                       "repair_frames": vstats["repair_frames"], "repair_steps": vstats["repair_steps"], "runners_to_chunk_end": vstats["runners_to_chunk_end"],
                       "replay_frames": vstats["replay_frames"], "replay_calls": vstats["replay_calls"], "runners_past_replay": vstats["runners_past_replay"], "retained_steps": n_retained,
                       "driver": ("C++ (sind_seq_*, csrc/host/seq.cpp)" + ("; fingerprints and seam states between the ranks over TCP (sind_seq_net_tcp)" if world > 1 else "")) if use_cabi else "Python (sindslam_amd/sequence.py VerifiedChunks)",
                       "max_frames_to_converge": vstats["max_frames_to_converge"], "repair_seconds": vstats["repair_seconds"], "lockstep_seconds": t_main,
                       "repair_pipeline": f"{R} runners x {Tr} frames per step" if verify else None,
                       "note": "every chunk seam is verified by comparing 128-bit fingerprints of the inter-frame state (the chunk's rebuilt state vs the predecessor's true end state); "
                               "a mismatching chunk is re-run from the true state until its state equals the speculative one of the same frame -- all inside the timed region; "
                               "replay_frames were re-run as tails only on retained phase-A outputs, repair_frames as whole frames on the repair pipeline"},
            "repaired_chunks": vstats["repaired_chunks"], "repair_frames": vstats["repair_frames"] + vstats["replay_frames"],
            "note": "value = owned frames of the whole sequence / wall time of ALL the work (the state warm-up frames of every chunk after the first, the seam verification "
                    "and the repair runs are inside the timed region); value_excl_warmup counts every processed frame as if it were owned (the pipeline's processing rate)",
            "mask_gather": ("%s all_gather per step, %.1f MB per rank" % ("RCCL through the C ABI (sind_pipe_gather_masks)" if args.collective == "cabi" else "RCCL" if args.backend == "nccl" else args.backend, S * T * H * W / 1e6)) if pg else "single rank (no collective)",
            "sequence_masks_bytes_per_rank": int(seq_masks.numel())}
    if cabi is not None:
        cabi.close()
    if use_cabi:
        job.close()
        if net is not None:
            net.close()
    else:
        pipe.close()
        if rp is not None:
            rp.close()
    # ---- rank 0: the first frames again in the in-order ("exact") mode -> IoU of the chunked masks at and behind the chunk seams, and the exact mode's own rate
    if rank == 0 and exact_leg and n > 1:
        from sindslam_amd.pipeline import Pipeline
        P = plan.processed
        E = min(plan.frames, args.exact_leg_frames) if args.exact_leg_frames > 0 else min(max(320, P + 2 * (P - SW)), plan.frames, 480)      # chunk 0, chunks 1-2 and what else fits: long enough for a steady-state rate
        Te = 32 if E >= 64 else 16
        nst = -(-E // Te); E = nst * Te
        ex = make_pipeline(cfg, intr, 1, Te, local)
        ex.prime(0, base_b[fidx(-1)], base_b[fidx(-2)]); ex.set_depth_ahead(True)
        eb = [bb[torch.tensor([fidx(k * Te + t) for t in range(Te)], device="cuda")].contiguous() for k in range(nst)]
        ed = [bd[torch.tensor([fidx(k * Te + t) for t in range(Te)], device="cuda")].contiguous() for k in range(nst)]
        torch.cuda.synchronize(); exact = np.zeros((E, H, W), np.uint8); te0 = time.perf_counter(); prev = None
        for k in range(nst):
            if ex.submit_dev(eb[k].data_ptr(), ed[k].data_ptr()):
                exact[prev * Te:(prev + 1) * Te] = ex.dyna[0]
            prev = k
        if ex.flush():
            exact[prev * Te:(prev + 1) * Te] = ex.dyna[0]
        te = time.perf_counter() - te0; ex.close()
        sm = seq_masks.reshape(n, P, H, W)
        ious = []; seam = []; inter_sum = union_sum = 0; equal_frames = 0
        for g, c in enumerate(plan.chunks):
            if g == 0:
                continue                                          # chunk 0 starts like the sequential run: identical by construction
            for q in range(c.first, min(c.last, E)):
                raw_ = sm[g, q - c.start].cpu().numpy(); equal_frames += int(np.array_equal(raw_, exact[q]))
                a_ = raw_ == 255; b_ = exact[q] == 255; u = np.logical_or(a_, b_).sum()
                it_ = np.logical_and(a_, b_).sum(); inter_sum += int(it_); union_sum += int(u)
                v = 1.0 if u == 0 else float(it_ / u); ious.append(v)
                if q == c.first: seam.append(v)
        info.update({"seam_iou_mean": float(np.mean(ious)) if ious else None, "seam_iou_min": float(np.min(ious)) if ious else None,
                     "seam_iou_pooled": (inter_sum / union_sum) if union_sum else None, "seam_iou_below_0.99": int(sum(v < 0.99 for v in ious)),
                     "seam_iou_first_frames": seam[:8], "seam_frames_compared": len(ious), "seam_masks_equal": equal_frames,
                     "seam_note": "chunked mode vs the in-order run on the same GPU code, owned frames of the chunks after the first: seam_masks_equal counts the frames whose "
                                  "imgDyna is byte-identical (with verification: all of them)",
                     "exact_mode": {"frames": E, "frames_per_step": Te, "fps": E / te,
                                    "bound": "1 / per-frame latency of the slower tail chain (depth chain: k-means warm labels; flow chain: weights, previous high mask); "
                                             "host + launch latency bound, does not grow with the number of GPUs"}})
    del seq_masks, dev_b, dev_d, bb, bd
    return dt, acc, info, loads


# ------------------------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    # HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The pipeline drives ~45 streams per handle (three flow slices, ORB, one per pool
    # worker, k-means, CalOccluded, region grow) and a sequence job carries a second, small handle for the repair runs: with four queues the flow slices of a
    # stand-alone sequence job share queues with tail streams and run 17 % slower than the same job after a streams run in the same process (250 vs 208 ms of dense
    # flow per step); with six or eight the two agree and the streams headline gains 3 %; ten and more are catastrophic (profiles/r04/hw_queues.txt).  Read by the runtime when it starts: set before torch / HIP load.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
    args.pipelined = not args.sync and not args.host_input
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE JSON line on stdout: libraries write banners there (RCCL prints its version block when a communicator is created), so file descriptor 1 is
    # pointed at stderr for the duration of the run and the line goes out through the saved descriptor
    sys.stdout.flush(); real_stdout = os.dup(1); os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    if args.rendezvous_only:                 # launcher test: meet, count, leave (no GPU needed)
        import torch
        import torch.distributed as dist
        loads = [host_load(time.process_time(), time.perf_counter() - 1e-3, cgroup_throttle())]
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            t = torch.ones(1); dist.all_reduce(t); seen = int(t.item())
            loads = gather_host_load(loads[0], True, "cpu"); dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            emit({"ranks_seen": seen, "n_gpus": world, "gpus_arg": args.gpus, "warmup": args.warmup, "steps": args.steps, **host_load_fields(loads)})
        return

    import numpy as np
    cfg = dict(CONFIGS[args.config])
    if args.flow_levels is not None:
        cfg["flow_max_levels"] = args.flow_levels
    cfg["flow_slices"] = args.flow_slices; cfg["flow_opts_off"] = args.flow_opts_off
    workload = args.workload if args.workload != "auto" else ("sequence" if world > 1 else "streams")
    T, K, Wm = args.frames_per_step, args.steps, args.warmup
    S = args.streams or (cfg["streams"] if workload == "streams" else sequence_streams(world, args.sequence_frames, cfg["streams"], K, args.seq_warmup_frames))
    seq_leg = workload == "streams" and world == 1 and not args.no_sequence_leg and not args.host_input       # N = 1 point of the sequence curve next to the streams headline
    import sindslam_amd.synth as SY
    intr0 = getattr(SY, cfg["intr"]); sc = cfg["width"] / 640.0
    intr = dict(intr0, fx=intr0["fx"] * sc, fy=intr0["fy"] * sc, cx=intr0["cx"] * sc, cy=intr0["cy"] * sc)
    # ---- synthetic input on the host, before torch / HIP are loaded (the frame generator forks workers)
    seq_b = seq_d = None
    if workload == "sequence" or seq_leg:
        seq_b, seq_d = base_frames(cfg, SEQ_BASE_FRAMES, 12345)          # every rank builds the same sequence
    if workload == "streams":
        nsteps = K + Wm
        ndata = min(nsteps, 12)             # distinct steps of input kept in host + device memory; longer runs cycle through them (the jump at the wrap is just another large-motion pair)
        base_b, base_d = base_frames(cfg, T * ndata + 2, 12345 + rank, nseeds=3)      # three scenes; the first eight streams (the parity sample) cover all of them
        bgr, depth = stream_variants(base_b, base_d, S)

        def frames_of_stream(s, count):
            return bgr[s, :count], depth[s, :count]

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    local = local % torch.cuda.device_count()          # rehearsal on a 1-GPU box: ranks share the card
    torch.cuda.set_device(local)
    # a process group exists for several ranks, and for ONE rank when --collective-at-1 asks for it (the RCCL gather path on a single card: same calls,
    # same buffers, world size 1 -- what can be rehearsed of the multi-GPU path on a one-GPU box)
    pg = world > 1 or args.collective_at_1
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 2000))
        dist.init_process_group(args.backend, rank=rank, world_size=world)     # backend "nccl" is RCCL on ROCm
    comm_dev = "cuda" if (pg and args.backend == "nccl") else "cpu"
    H, W = cfg["height"], cfg["width"]
    seq_info = None; tum_info = None; first_dyna = first_kps = None; dropin_info = small_info = None

    if workload == "sequence":
        if os.environ.get("SIND_BENCH_PRE"):          # experiment: a pipeline created and destroyed before the job (what the one-GPU line's streams workload leaves behind)
            ps, pt = [int(x) for x in os.environ["SIND_BENCH_PRE"].split("x")]
            pre = make_pipeline(cfg, intr, ps, pt, local, args.host_threads); pre.close(); del pre
        dt, acc, seq_info, loads = sequence_job(args, cfg, intr, seq_b, seq_d, rank, world, local, pg, comm_dev, S, K, Wm, not args.no_exact_leg)
        T = seq_info["frames_per_step_per_chunk"]; pairs = seq_info["owned_frames"]; flush_ms = seq_info["final_flush_ms"]; loop_ms = sync_ms = None
        seq_info["pairs_per_rank_step"] = S * T
        if world > 1 and not args.no_n1_leg:
            # the N = 1 point of the strong-scaling curve in the same line: rank 0 runs the SAME job alone on its GPU after the timed region, the others wait
            if rank == 0:
                torch.cuda.empty_cache()
                S1 = sequence_streams(1, args.sequence_frames, cfg["streams"], K, args.seq_warmup_frames)
                d1, _, i1, _ = sequence_job(args, cfg, intr, seq_b, seq_d, 0, 1, local, False, "cpu", S1, K, min(Wm, 2), False)
                seq_info["n1_value"] = i1["value"]; seq_info["n1_seconds"] = d1; seq_info["n1_chunks"] = i1["chunks"]; seq_info["n1_verify"] = i1["verify"]
                seq_info["speedup_vs_n1"] = seq_info["value"] / i1["value"]; seq_info["scaling_efficiency"] = seq_info["value"] / (world * i1["value"])
                seq_info["n1_note"] = "the same fixed-length job on rank 0's GPU alone, run after the timed region of the N-rank job (same code path as --gpus 1's `sequence` leg)"
                # what ONE GPU does at THIS job's step size (chunks per rank x frames per step): efficiency lost to the step size and efficiency lost to the
                # exchange between the ranks can then be told apart -- value / (world * n1_small_step_value) is the share the communication and the seams cost
                torch.cuda.empty_cache()
                vb, vd = stream_variants(seq_b, seq_d, S)
                ss = small_step_leg(cfg, intr, vb, vd, local, args.host_threads, S2=S, T2=T, steps=max(4, min(K, (SEQ_BASE_FRAMES - 2) // max(T, 1))), warm=2)
                seq_info["n1_small_step_value"] = ss["value"]; seq_info["n1_small_step"] = ss
            dist.barrier()
        ranks_seen = 1
        if pg:
            rs = torch.ones(1, device=comm_dev); dist.all_reduce(rs); ranks_seen = int(rs.item())
    else:
        from sindslam_amd.parallel import gather_masks
        parts = pipelines_for(S, T, args.pipelines)
        pipe = make_pipeline(cfg, intr, S, T, local, args.host_threads, parts)
        # inputs resident in HBM before the timed region, laid out [step][S][T]...
        for s in range(S):
            pipe.prime(s, bgr[s, 1], bgr[s, 0])
        dev_b = [torch.from_numpy(np.ascontiguousarray(bgr[:, 2 + i * T: 2 + (i + 1) * T])).cuda() for i in range(ndata)]
        dev_d = [torch.from_numpy(np.ascontiguousarray(depth[:, 2 + i * T: 2 + (i + 1) * T]).view(np.int16)).cuda() for i in range(ndata)]
        torch.cuda.synchronize()
        gbuf = {}

        def gather():
            """several ranks (stream-sharded replicas): RCCL all_gather of this step's per-frame dynamic masks (page-locked source, persistent device buffers)"""
            if not pg:
                return
            m = pipe.dyna_pinned if pipe.dyna_pinned is not None else torch.from_numpy(pipe.dyna)
            if comm_dev == "cuda":
                if "dev" not in gbuf:
                    gbuf["dev"] = torch.empty_like(m, device="cuda"); gbuf["out"] = torch.empty((world,) + tuple(m.shape), dtype=m.dtype, device="cuda")
                gbuf["dev"].copy_(m, non_blocking=True)
                gather_masks(gbuf["dev"], out=gbuf["out"])
                torch.cuda.current_stream().synchronize()
            else:
                gather_masks(m)

        NPS = min(S, args.cpu_threads) if (world == 1 and not args.no_cpu_baseline) else 0

        def parity_sample():
            return [pipe.dyna[s].copy() for s in range(NPS)], [[pipe.keypoints(s, t)[0].copy() for t in range(T)] for s in range(NPS)]

        host_b = host_d = None
        if args.host_input:
            host_b = [t_.cpu().numpy() for t_ in dev_b]; host_d = [t_.cpu().numpy().view(np.uint16) for t_ in dev_d]
        for i in range(Wm):                     # warm-up: synchronous steps
            pipe.process_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()); gather()
            if i == 0:                          # the first streams' first T results, for the parity figures below
                first_dyna, first_kps = parity_sample()
        if pg:
            dist.barrier()
        torch.cuda.synchronize()
        thr0 = cgroup_throttle()
        t0 = time.perf_counter(); c0 = time.process_time(); th0 = thread_cpu_seconds() if args.thread_cpu else None
        acc = StepAcc(); have_pending = False
        for i in range(Wm, Wm + K):             # timed steps
            if args.host_input:
                pipe.process(host_b[i % ndata], host_d[i % ndata]); gather()
            elif not args.pipelined:
                pipe.process_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()); gather()
            else:                               # software-pipelined: phase A of step i overlaps the tails of step i-1, whose results arrive now
                _ts = time.perf_counter(); _hv = pipe.submit_dev(dev_b[i % ndata].data_ptr(), dev_d[i % ndata].data_ptr()); acc.submit_wall += time.perf_counter() - _ts
                if _hv:
                    gather()
                have_pending = True
            if first_dyna is None and not args.pipelined:     # --warmup 0: take the parity sample from the first timed step (a few MB copied)
                first_dyna, first_kps = parity_sample()
            acc.add(pipe.stats(), args.pipelined)
        t_flush0 = time.perf_counter()
        if args.pipelined and have_pending and pipe.flush():      # drain the last step inside the timed region
            gather()
        flush_ms = (time.perf_counter() - t_flush0) * 1e3; loop_ms = (t_flush0 - t0) * 1e3
        _tsy = time.perf_counter(); torch.cuda.synchronize(); sync_ms = (time.perf_counter() - _tsy) * 1e3
        if pg:
            dist.barrier()
        th1 = thread_cpu_seconds() if args.thread_cpu else None
        load = host_load(c0, t0, thr0) + host_info_list(pipe); dt = time.perf_counter() - t0
        ranks_seen = 1
        if pg:
            tt = torch.tensor([dt], device=comm_dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
            rs = torch.ones(1, device=comm_dev); dist.all_reduce(rs); ranks_seen = int(rs.item())
        loads = gather_host_load(load, pg, comm_dev)
        pairs = S * T * K * world
        if args.thread_cpu and rank == 0:       # short-lived threads (flow slices, ORB, octree) that ended before the second sample are not listed
            cpu_s = load[0] * (time.perf_counter() - t0)
            for name in sorted(th1, key=lambda n_: -(th1[n_] - th0.get(n_, 0.0))):
                print(f"[thread-cpu] {name:16s} {(th1[name] - th0.get(name, 0.0)) / K * 1e3:9.1f} core-ms per step", file=sys.stderr)
            live = sum(th1[n_] - th0.get(n_, 0.0) for n_ in th1)
            print(f"[thread-cpu] {'(exited threads)':16s} {(cpu_s - live) / K * 1e3:9.1f} core-ms per step   total {cpu_s / K * 1e3:.1f}", file=sys.stderr)
        grow_q = pipe.grow_share(); km_groups = pipe.kmeans_groups()
        pipe.close(); del dev_b, dev_d, pipe
        dropin_info = small_info = None
        if world == 1 and not args.host_input:
            torch.cuda.empty_cache()
            if not args.no_small_step_leg and S >= 16:
                small_info = small_step_leg(cfg, intr, bgr, depth, local, args.host_threads)
            if not args.no_dropin_leg:
                dropin_info = dropin_leg(cfg, intr, bgr[0], depth[0], local, args.dropin_frames)
        # ---- one GPU: the fixed-length sequence job as well (the N = 1 point of the curve that --gpus N > 1 reports as its headline)
        if seq_leg:
            torch.cuda.empty_cache()
            Ss = sequence_streams(1, args.sequence_frames, cfg["streams"], max(K, 10), args.seq_warmup_frames)
            sdt, sacc, seq_info, _ = sequence_job(args, cfg, intr, seq_b, seq_d, 0, 1, local, False, "cpu", Ss, max(K, 10), min(Wm, 2), not args.no_exact_leg)
            seq_info["solver_busy_ms_per_step"] = sacc.sor_union / max(K, 10); seq_info["ms_per_step"] = sdt / max(K, 10) * 1e3
            seq_info["leg_note"] = "same job as the headline of --gpus N > 1 (strong scaling over the ranks), run here after the timed region of the streams workload"
            if not args.no_tum_leg:
                # BASELINE.json configs[1] read literally: ONE sequence of TUM fr3/walking_xyz's length (the committed sample trajectory has 823 poses) on one GPU,
                # every frame equal to the sequential loop (verified chunks), checked against the in-order run over the WHOLE sequence
                import copy
                a2 = copy.copy(args); a2.sequence_frames = TUM_SEQUENCE_FRAMES; a2.exact_leg_frames = TUM_SEQUENCE_FRAMES
                torch.cuda.empty_cache()
                K2, S2 = 6, 14               # measured (profiles/r04/tum_length_chunking.txt): 14 chunks x 13 frames per step; more chunks pay more warm-up, fewer make the tail chain the step time
                tdt, _, tum_info, _ = sequence_job(a2, cfg, intr, seq_b, seq_d, 0, 1, local, False, "cpu", S2, K2, 1, not args.no_exact_leg)
                tum_info["leg_note"] = ("one sequence of %d frames (the length of TUM fr3/walking_xyz) on this GPU in %d steps: frames/s with verification and repairs inside the clock; "
                                        "seam_masks_equal counts the frames byte-identical to the in-order run over the whole sequence" % (TUM_SEQUENCE_FRAMES, K2))

    if rank == 0:
        seq = workload == "sequence"
        out = {
            "metric": "DynaDetect+ORB frame-pairs/sec at 640x480; mask IoU vs CPU ref", "value": pairs / dt, "unit": "frame-pairs/s",
            "n_gpus": ranks_seen, "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if seq else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"] + ("; ONE sequence of %d frames, frame-sharded in %d lock-step chunks (24 state warm-up frames per chunk inside the timed region), per-step %s gather of the dynamic masks"
                                                      % (args.sequence_frames, world * S, "RCCL" if args.backend == "nccl" else args.backend) if seq else ""),
                       "name": args.config, "mode": workload, "streams_per_gpu": S, "frames_per_step": T, "frame_pairs_per_step": S * T * world,
                       "parallelism": ("frame-sharded x%d" if seq else "stream-sharded x%d") % world, "pipelined": bool(args.pipelined or seq),
                       "pipelines_per_gpu": (seq_info.get("pipelines_per_gpu") if seq else parts),
                       "flow_pyramid_levels": cfg["flow_max_levels"] or "all (49 at 640x480)",
                       "inputs": "host buffers, H2D inside the timed region" if args.host_input else "resident in HBM"},
            "ranks_seen": ranks_seen,
            "roofline": roofline_of(acc, K, dt, S * T / max(acc.sor_slices, 1), args.config, cfg),
            "stage_ms_per_step": {"front": acc.stages[0] / K, "dense_flow": acc.stages[1] / K, "orb_front": acc.stages[2] / K, "tails": acc.stages[3] / K, "host_upload": acc.stages[5] / K, "total": acc.stages[4] / K,
                                  "tails_wait_after_phase_a": acc.tail_wait / K, "final_flush_total": flush_ms, "submit_call_wall": acc.submit_wall * 1e3 / K, "loop_total": loop_ms, "final_sync": sync_ms},
        }
        out.update(host_load_fields(loads))
        # PEAC region grow of CalOccluded: share of the frames grown on the GPU at the end of the run, in quarters (adaptive by default; same results either way)
        out["region_grow_gpu_quarters"] = grow_q if not seq else seq_info.get("region_grow_gpu_quarters")
        # groups of streams whose batched k-means rounds run as independent chains at the end of the run (2..4, same controller; same results)
        out["kmeans_groups"] = km_groups if not seq else seq_info.get("kmeans_groups")
        if not seq and dropin_info:
            out["dropin"] = dropin_info
        if not seq and small_info:
            out["small_step"] = small_info
        if seq_info:
            out["sequence"] = seq_info
        if tum_info:
            out["sequence_tum_length"] = tum_info
        if not args.no_cpu_baseline and world == 1 and first_dyna is not None:
            out["cpu_baseline"], out["parity"] = cpu_baseline(frames_of_stream, intr, cfg, min(T, 4), first_dyna, first_kps, threads=args.cpu_threads, n_streams=NPS)
        else:
            out["cpu_baseline"] = None
        emit(out)
    if pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
