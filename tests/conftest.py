import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """Create torch's HIP context before any test runs for minutes on the CPU side: on the GPU boxes a torch.cuda
    initialisation late in a long session has failed with 'No HIP GPUs are available' while libagz kept working."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
