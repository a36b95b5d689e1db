import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rtl-ws_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Product libraries present (built in-tree by __graft_entry__.build())."""
    import rtlws
    if not (os.path.exists(rtlws.HIP_LIB) and os.path.exists(rtlws.AMD_LIB)):
        rtlws.build()
    return rtlws


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def engine(built):
    """HIP engine on device 0; fails (not skips) when the extension or the
    device is missing so a silent fallback can never pass a gpu test."""
    eng = built.Engine(0)
    yield eng
    eng.close()


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
