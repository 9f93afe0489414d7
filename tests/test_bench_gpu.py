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
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["MASTER_PORT"] = str(29400 + os.getpid() % 1500)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--streams", "4",
                        "--frames-per-step", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "spawned 2 ranks" in r.stderr and "torch imported in the launcher: False" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0]); s = d["sequence"]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["mode"] == "sequence" and d["config"]["parallelism"] == "frame-sharded x2"
    assert d["value"] > 0 and d["roofline"]["launches"] > 0 and d["warmup"] >= 6                       # at least 24 state warm-up frames per chunk
    assert s["chunks"] == 8 and s["chunk_frames"] == 8 and s["owned_frames"] == 64 and s["frames"] == 2 + 64 + d["warmup"] * 4
    assert s["seam_frames_compared"] > 0 and s["seam_iou_mean"] >= 0.95 and s["exact_mode"]["fps"] > 0
