"""GPU: the C-ABI collective (sind_comm_*, sind_pipe_gather_masks): RCCL all-gather of a step's dynamic masks without torch.distributed.  World size 1 on
the one card of the box; two ranks need two GPUs (RCCL refuses two ranks on one device) and are skipped cleanly otherwise."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_of_one_gathers_its_own_block():
    from sindslam_amd.parallel import Comm
    uid = Comm.unique_id(); assert len(uid) == 128
    c = Comm(uid, 0, 1, 0)
    x = np.random.default_rng(1).integers(0, 256, (3, 2, 48, 64), dtype=np.uint8)
    dev, host = c.allgather(x)
    assert dev.shape == (1, 3, 2, 48, 64) and np.array_equal(dev.cpu().numpy()[0], x) and np.array_equal(host[0], x)
    c.close()


def test_pipeline_masks_through_the_c_abi_gather(frames):
    from sindslam_amd.parallel import Comm
    from sindslam_amd.pipeline import Pipeline
    from sindslam_amd.synth import TUM3
    bgr, depth = frames
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    pipe = Pipeline(1, 2, 640, 480, *K, 1500, 1.2, 8, 15, 5)
    pipe.prime(0, bgr[1], bgr[0]); pipe.process(bgr[None, 2:4], depth[None, 2:4])
    c = Comm(Comm.unique_id(), 0, 1, 0)
    g = c.gather_pipeline_masks(pipe)
    assert g.shape == (1, 1, 2, 480, 640) and np.array_equal(g.cpu().numpy()[0], pipe.dyna) and (pipe.dyna == 255).any()
    c.close(); pipe.close()


_WORKER = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from sindslam_amd.parallel import Comm
rank, world, uid = int(sys.argv[2]), int(sys.argv[3]), bytes.fromhex(sys.argv[4])
try:
    c = Comm(uid, rank, world, rank if int(sys.argv[5]) > 1 else 0)
except Exception as e:
    print("CREATE_FAILED", e); sys.exit(3)
x = np.full((4, 8), 10 + rank, np.uint8)
dev, host = c.allgather(x)
assert host[:, 0, 0].tolist() == [10, 11], host[:, 0, 0]
print("OK"); c.close()
'''


@pytest.mark.timeout(300)
def test_two_ranks_when_two_gpus_are_there():
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("one GPU on this box: RCCL does not place two ranks of a communicator on one device")
    from sindslam_amd.parallel import Comm
    uid = Comm.unique_id().hex()
    ps = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, str(r), "2", uid, str(ngpu)], stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in ps]
    assert all("OK" in o for o in outs), outs
