import os
import sys

import pytest

try:
    # torch's libamdhip64 must be the one HIP runtime of the process: load it before libbeifong_hip.so (a test file that
    # touches the product library first and torch later otherwise sees "No HIP GPUs are available")
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hiplib():
    """The product library; GPU tests must fail loudly if it is missing."""
    from beifong_amd import capi
    return capi.load_library()
