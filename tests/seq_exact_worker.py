"""worker of tests/test_sequence_gpu.py::test_exact_sequence_two_ranks_hand_the_state_over: one rank of the in-order sequence mode"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    n, out_dir = int(sys.argv[1]), sys.argv[2]
    import torch.distributed as dist
    from sindslam_amd.sequence import process_sequence_exact
    from sindslam_amd.synth import SyntheticStream, TUM3
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bgr, depth = SyntheticStream(seed=99).frames(0, n)
    got = process_sequence_exact(bgr, depth, TUM3, frames_per_step=6, rank=rank, world=world, want_keypoints=False, first_step=3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), dyna=got["dyna"], label=got["label"], owned=np.array(got["owned"]))
    dist.destroy_process_group()
