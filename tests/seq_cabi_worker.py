"""worker of tests/test_sequence_gpu.py::test_cabi_ranks_forced_mismatch_over_tcp: one rank of the chunked sequence mode through the C ABI (sind_seq_*), the exchange
between the ranks over loopback TCP; the ranks share the card.  No torch.distributed."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    n, out_dir, warmup, rank, world, port = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    from sindslam_amd.seq import SeqNet, run_sequence
    from sindslam_amd.synth import SyntheticStream, TUM3
    bgr, depth = SyntheticStream(seed=99).frames(0, n)
    net = SeqNet.tcp(rank, world, port)
    st = {}
    got = run_sequence(bgr, depth, TUM3, streams=2, frames_per_step=3, warmup=warmup, net=net, want_keypoints=False, repair_streams=2, repair_frames_per_step=3, stats=st)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), dyna=got["dyna"], label=got["label"], mask=got["mask"], owned=np.array(got["owned"]), stats=json.dumps(st))
    net.close()
