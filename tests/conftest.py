import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import helpers as H
    return np.load(os.path.join(H.GOLDEN, "hotpath.npz"))


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """torch must initialise HIP before libsfmx does in the same process (otherwise torch reports
    'No HIP GPUs are available'); harmless on the CPU-only container."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
    yield
