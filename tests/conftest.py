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
    need = [rtlws.HIP_LIB, rtlws.AMD_LIB] + [os.path.join(rtlws.LIB_DIR, f) for f in
                                             ("librtlws_cbb.so", "librtlws_synth.so", "rtlws_dropin_demo", "rtlws_multi_stream")]
    if not all(os.path.exists(f) for f in need):
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
