"""worker of tests/test_sequence_gpu.py::test_verified_chunks_two_ranks_forced_mismatch: one rank of the chunked sequence mode (gloo; both ranks share the card)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    n, out_dir, warmup = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    import torch.distributed as dist
    from sindslam_amd.sequence import process_sequence
    from sindslam_amd.synth import SyntheticStream, TUM3
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bgr, depth = SyntheticStream(seed=99).frames(0, n)
    st = {}
    got = process_sequence(bgr, depth, TUM3, streams=2, frames_per_step=3, warmup=warmup, rank=rank, world=world, want_keypoints=False, repair_streams=2,
                           repair_frames_per_step=3, stats=st)
    st.pop("plan", None)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), dyna=got["dyna"], label=got["label"], mask=got["mask"], owned=np.array(got["owned"]), stats=json.dumps(st))
    dist.destroy_process_group()
