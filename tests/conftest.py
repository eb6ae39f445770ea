import os
import sys

import pytest

# torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first in a process is the one
# every later library binds to (same SONAME).  Import torch before any test module dlopens libsfmx*.so.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

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
