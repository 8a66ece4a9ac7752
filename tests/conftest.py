import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture()
def oracle_backend():
    """CPU execution of the HIP-backed operators through the test oracle (never in product code)."""
    from oracle import torch_backend
    torch_backend.install()
    yield torch_backend
    torch_backend.uninstall()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran without a GPU; run the CPU suite with -m 'not gpu'")
    return torch.device("cuda:0")
