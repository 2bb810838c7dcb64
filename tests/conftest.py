import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_lib():
    """libcozo_gpu with a device selected; GPU tests fail loudly if the HIP library or the device is missing."""
    from cozo_amd import _lib
    L = _lib.lib()
    rc = L.cz_init(0)
    assert rc == 0, L.cz_last_error()
    return L
