import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle.backend import OracleBackend, build
    build()
    return OracleBackend()


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def measure_backend(hip_device):
    """HIP backend bound to the MEASUREMENT build of the library (-DRP_MEASUREMENT: the rp_set_* knobs and the kernel
    instances they select; its own handle next to the product library).  Tests use it to show that a knob changes how a
    kernel walks its operands, never the bits."""
    from rectipy_amd import build
    from rectipy_amd.engine import HipBackend
    build.build(measurement=True)
    return HipBackend(measurement=True)
