import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (dev container)")


@pytest.fixture(scope="session")
def kartohip_lib():
    """Builds (if hipcc is present and sources changed) and loads libkartohip.so."""
    from slam_toolbox_amd import build, capi
    try:
        build.build()
    except Exception:
        if not os.path.exists(capi.LIB_PATH):
            raise
    return capi.lib()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import karto
    return karto.lib()
