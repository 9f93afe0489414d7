import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def stream():
    from sindslam_amd.synth import SyntheticStream
    return SyntheticStream()


@pytest.fixture(scope="session")
def frames(stream):
    return stream.frames(0, 6)
