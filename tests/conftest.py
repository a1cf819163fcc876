import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (checker only), built on demand from oracle/gpx_oracle.cpp."""
    from tests.oracle_binding import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def hip_lib():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.zeros(1, device="cuda:0")  # wake the device before the HIP library's own runtime looks for it
    torch.cuda.synchronize()
    from gigapaxos_amd import load_hip

    return load_hip()
