import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_ext():
    from oracle.pointnet2_oracle import ext

    return ext


@pytest.fixture(scope="session")
def hip_ext():
    """The product `_ext` (HIP). Fails loudly if the library is missing."""
    import torch

    assert torch.cuda.is_available(), "gpu test without a GPU"
    from unopose_amd.pointnet2 import _ext

    return _ext
