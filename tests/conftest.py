import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # fresh checkout: build the product (hipcc cross-compiles gfx950 without a GPU), its host-stage shim and the oracle once
    need = [os.path.join(ROOT, "sindslam_amd", "libsind_hip.so"), os.path.join(ROOT, "sindslam_amd", "libsind_host.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def stream():
    from sindslam_amd.synth import SyntheticStream
    return SyntheticStream()


@pytest.fixture(scope="session")
def frames(stream):
    return stream.frames(0, 6)
