"""GPU: bench.py end to end in its multi-rank shape on this one card -- `--gpus 2` starts two ranks itself (gloo, so that both may share the GPU), which
run the frame-sharded sequence workload with the per-step mask gather, the in-order re-run of the first chunks and the one JSON line of rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_two_ranks_frame_sharded_sequence_line():
    """the fixed-length sequence job on two ranks: `warmup` is the argument, the chunks' state warm-up runs INSIDE the timed region (value counts owned
    frames only), and the seam check against the in-order run holds the bar the bench reports"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["MASTER_PORT"] = str(29400 + os.getpid() % 1500)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--streams", "2",
                        "--sequence-frames", "200", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "spawned 2 ranks" in r.stderr and "torch imported in the launcher: False" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0]); s = d["sequence"]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["mode"] == "sequence" and d["config"]["parallelism"] == "frame-sharded x2"
    assert d["warmup"] == 1 and d["steps"] == 4 and d["scaling"] == "strong"
    # 200 frames on 4 chunks in 4 steps: P >= (200 + 3 * 16) / 4 = 62 -> T = 16, every chunk processes 64 frames, chunks 1.. own 48 each
    assert s["chunks"] == 4 and s["frames_per_step_per_chunk"] == 16 and s["processed_frames_per_chunk"] == 64 and s["owned_frames"] == 200 and s["processed_frames"] == 256
    assert s["state_warmup_frames"] == 16 and s["state_warmup_steps"] == 1
    assert d["value"] > 0 and abs(d["value"] - s["value"]) < 1e-6 * d["value"] and s["value"] <= s["value_excl_warmup"]
    assert abs(s["value_excl_warmup"] / s["value"] - 256 / 200) < 1e-6
    rf = d["roofline"]       # (two chunks per rank: no streaming launches; the tiled / one-workgroup launches are counted beside them)
    assert rf["bound"] == "hbm" and rf["launches"] + rf["other_solver_kernels"]["launches"] > 0 and (rf["frac"] is None or 0 < rf["frac"] <= 1.0)
    assert len(d["host_by_rank"]) == 2 and all(h["host_cores_busy"] > 0 for h in d["host_by_rank"])
    # chunked masks vs the in-order run of the same frames (owned frames of chunks 1..3 inside the first E frames)
    # verified chunks: every compared frame is byte-identical to the in-order run, whatever the seams needed (verification and repairs are inside the clock)
    assert s["seam_frames_compared"] >= 100 and s["exact_mode"]["fps"] > 0 and s["exact"] is True
    assert s["seam_masks_equal"] == s["seam_frames_compared"] and s["seam_iou_min"] == 1.0 and s["seam_iou_below_0.99"] == 0
    v = s["verify"]; assert v["seams"] == 3 and 0 <= v["mismatched_seams"] <= 3 and (v["repair_frames"] + v["replay_frames"] > 0) == (v["mismatched_seams"] > 0)
    # the line carries its own N = 1 point (same job on rank 0 alone, after the timed region)
    assert s["n1_value"] > 0 and abs(s["speedup_vs_n1"] - s["value"] / s["n1_value"]) < 1e-9 and abs(s["scaling_efficiency"] - s["speedup_vs_n1"] / 2) < 1e-9
    # ... and the one-GPU rate AT THIS JOB'S STEP SIZE, so that efficiency lost to the step size and to the exchange can be told apart
    assert s["pairs_per_rank_step"] == 2 * 16 and s["n1_small_step_value"] > 0 and s["n1_small_step"]["pairs"] == s["pairs_per_rank_step"]


@pytest.mark.timeout(900)
def test_one_gpu_line_carries_the_sequence_leg():
    """--gpus 1: streams headline (weak) plus the fixed-length sequence job as `sequence` (the N = 1 point of the strong-scaling curve)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--streams", "8", "--sequence-frames", "300", "--no-cpu-baseline",
                        "--no-exact-leg"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0]); s = d["sequence"]
    assert d["config"]["mode"] == "streams" and d["scaling"] == "weak" and d["warmup"] == 1 and d["steps"] == 2 and d["config"]["frame_pairs_per_step"] == 32
    assert s["frames"] == 300 and s["owned_frames"] == 300 and s["steps"] == 10 and 0 < s["value"] <= s["value_excl_warmup"]
    t = d["sequence_tum_length"]; assert t["frames"] == 830 and t["owned_frames"] == 830 and t["steps"] == 6 and t["exact"] is True and t["value"] > 0
    # the roofline object says one thing: the HBM roof of the streaming solver (no field above its peak); 32 pairs per step never reach that kernel, so frac is absent here
    r = d["roofline"]; assert r["bound"] == "hbm" and r["peak"] == 8000.0 and (r["frac"] is None or (0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9))
    assert r["valu_peak_ops_per_s"] == 78.6e12 and r["useful_ops_per_update"] == 32 and r["other_solver_kernels"]["launches"] > 0
    # the reference's call pattern, one frame per call with host pointers
    dr = d["dropin"]; assert dr["frames"] >= 100 and dr["fps"] > 0 and abs(dr["fps"] * dr["ms_per_frame"] - 1000.0) < 1e-6 and dr["detect_stages_ms"]["calls"] == dr["frames"]
    assert d.get("parity") is None or "sample_from" in d["parity"]


@pytest.mark.timeout(1200)
def test_sequence_at_the_drivers_chunk_count_every_seam_equal():
    """the sequence workload with the chunk count of the driver's one-GPU run (26 chunks, 20 steps) on a shorter sequence, and the in-order re-run over the WHOLE
    sequence: all 25 seams are covered, every owned frame of every later chunk is byte-identical to the in-order run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sequence", "--steps", "20", "--warmup", "1", "--streams", "26", "--sequence-frames", "1000",
                        "--exact-leg-frames", "1000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0]); s = d["sequence"]; v = s["verify"]
    print({k: s[k] for k in ("value", "value_excl_warmup", "chunks", "frames_per_step_per_chunk", "seam_frames_compared", "seam_masks_equal", "seam_iou_min")}, v)
    # 1000 frames on 26 chunks in 20 steps: T = 3, chunk 0 owns 60 frames, the others 44 -> 23 chunks own frames (22 seams), the last three run empty-handed
    assert s["chunks"] == 26 and s["steps"] == 20 and s["frames_per_step_per_chunk"] == 3 and v["seams"] == 22 and s["exact"] is True
    assert s["seam_frames_compared"] == 940 and s["seam_masks_equal"] == s["seam_frames_compared"] and s["seam_iou_min"] == 1.0
    assert v["repaired_chunks"] >= v["mismatched_seams"] and v["repair_seconds"] < s["seconds"]


@pytest.mark.timeout(900)
def test_three_ranks_short_warmup_repairs_across_rank_seams():
    """three ranks on this one card (gloo), 2 chunks each, a 2-frame warm-up: seams mismatch, the middle rank both receives its predecessor's end-state blob and sends its own,
    repaired masks travel in the per-round exchange -- and every compared frame equals the in-order run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["MASTER_PORT"] = str(27400 + os.getpid() % 1500)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--streams", "2",
                        "--sequence-frames", "150", "--seq-warmup-frames", "2", "--exact-leg-frames", "150", "--no-n1-leg", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0]); s = d["sequence"]; v = s["verify"]
    print({k: s[k] for k in ("value", "chunks", "frames_per_step_per_chunk", "seam_frames_compared", "seam_masks_equal")}, v)
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3 and s["chunks"] == 6 and s["exact"] is True and len(d["host_by_rank"]) == 3
    assert all(h["sizing"]["ranks_on_node"] == 3 and h["sizing"]["cpu_share"] >= 4 for h in d["host_by_rank"])
    assert s["seam_masks_equal"] == s["seam_frames_compared"] >= 100 and s["seam_iou_min"] == 1.0
