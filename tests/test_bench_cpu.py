"""CPU: bench.py's own launcher (--gpus N without torch.distributed.run) and its input helpers.  No GPU: the ranks only rendezvous."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_PORT"] = str(29400 + os.getpid() % 1500)
    return env


def test_gpus_flag_spawns_the_ranks_and_the_launcher_never_touches_torch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only", "--warmup", "5", "--steps", "20"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "spawned 2 ranks" in r.stderr and "torch imported in the launcher: False" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2
    assert out["warmup"] == 5 and out["steps"] == 20                 # the line repeats the arguments, whatever the workload does inside
    assert [h["rank"] for h in out["host_by_rank"]] == [0, 1]        # host load and CPU-quota counters of EVERY rank are gathered into the line
    assert all("host_cores_busy" in h and "cpu_quota" in h for h in out["host_by_rank"])


def test_under_a_launcher_no_second_set_of_ranks_is_started():
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "spawned" not in r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["ranks_seen"] == 1


def test_a_failing_rank_fails_the_launcher():
    """without a GPU every rank stops with 'needs an MI355X': the launcher must report that, not hang or print a result"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run the real bench")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--streams", "1"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "needs an MI355X" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_pingpong_sequence_and_frame_generator():
    sys.path.insert(0, ROOT)
    import bench
    P = 6
    idx = [bench.pingpong(f, P) for f in range(40)]
    assert idx[:11] == [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
    assert all(abs(a - b) == 1 for a, b in zip(idx, idx[1:]))             # consecutive frames are always neighbours of the base sequence
    cfg = dict(bench.CONFIGS["tum3"])
    b, d = bench.base_frames(cfg, 3, 12345)
    from sindslam_amd.synth import SyntheticStream
    rb, rd = SyntheticStream(seed=12345).frames(0, 3)
    assert np.array_equal(b, rb) and np.array_equal(d, rd)
    vb, vd = bench.stream_variants(b, d, 5)
    assert np.array_equal(vb[0], b) and np.array_equal(vb[1], b[:, :, ::-1]) and np.array_equal(vd[2], d[:, ::-1]) and vb[4].mean() < b.mean()
    b3, d3 = bench.base_frames(cfg, 2, 12345, nseeds=3)                   # three scenes
    assert b3.shape == (3, 2, 480, 640, 3) and np.array_equal(b3[0], b[:2]) and not np.array_equal(b3[1], b3[0])
    v3, _ = bench.stream_variants(b3, d3, 7)
    assert np.array_equal(v3[1], b3[1]) and np.array_equal(v3[3], b3[0][:, :, ::-1]) and np.array_equal(v3[5], b3[2][:, :, ::-1])


def test_sequence_chunk_count_heuristic():
    """chunks per GPU of the 4000-frame job at the driver's 20 steps: the chunk count shrinks as the ranks grow (the warm-up work per chunk is fixed)"""
    sys.path.insert(0, ROOT)
    import bench
    from sindslam_amd.sequence import plan_lockstep
    got = {n: bench.sequence_streams(n, 4000, 128, 20) for n in (1, 2, 4, 8)}
    assert got[1] > got[2] > got[4] > got[8] >= 4
    for n, s in got.items():
        p = plan_lockstep(4000, n * s, 20, bench.SEQ_WARMUP_FRAMES)
        assert p.processed_total <= 1.3 * 4000 and sum(c.last - c.first for c in p.chunks) == 4000
        if n == 8:
            assert p.T <= 4 and s * p.T <= 40          # a rank's ~30 pairs per step as at least eight short chains (the tails of a chunk's frames run one after the other)
