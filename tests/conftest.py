import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    return oracle.api()


@pytest.fixture(scope="session")
def gpu():
    """HIP engine function table; fails loudly (no skip, no fallback) when the device or library is missing."""
    import mvolps_amd

    mvolps_amd.require_device()
    return mvolps_amd.api()
