"""The strict-C99 reference-style caller (rtl-ws_amd/host/dropin_demo.c) runs
clean on the GPU: headers, linkage and return codes as a C maintainer sees them."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_dropin_demo_runs(built):
    exe = os.path.join(built.LIB_DIR, "rtlws_dropin_demo")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "peak_slot 612 expected 612" in out.stdout
    assert "dc_slot_equals_running_neighbour 1" in out.stdout
    assert "rf_decimator blocks 6" in out.stdout
