import os
import sys

import pytest

try:  # PyTorch-ROCm bundles its own HIP runtime: whichever runtime is loaded first owns the GPU for the process, so torch
    import torch  # noqa: F401  (must be imported before libslacken_amd.so is opened by any test that also uses torch tensors)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
