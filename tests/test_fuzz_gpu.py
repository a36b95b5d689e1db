"""One short seed of each randomised differential test on every `pytest -m gpu` run (the
fuzzers were run by hand).  tests/tools/fuzz_parity.py draws random descriptors and batch sizes for
rtlws_spectra_batch, tests/tools/fuzz_parity_f64.py for rtlws_spectra_batch_f64 -- both against the f64 oracle
(src/spectrum.c:15-99, src/cbb_main.c:106-135 restated); an assertion inside them exits non-zero.  The seed
changes with the date, so successive rounds walk different cases while a failure stays reproducible (the
seed is in the output)."""
import datetime
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["fuzz_parity.py", "fuzz_parity_f64.py"])
def test_short_fuzz_seed(built, script):
    seed = 1000 + datetime.date.today().toordinal() % 1000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", script), str(seed), "8"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, "seed %d\n%s\n%s" % (seed, r.stdout[-2000:], r.stderr[-4000:])
    assert "cases ok" in r.stdout and "seed %d" % seed in r.stdout, r.stdout
    cases = int(r.stdout.split("cases ok")[0].split()[-1])
    assert cases >= 5, r.stdout
