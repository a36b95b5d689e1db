"""One short seed of each randomised differential test on every `pytest -m gpu` run (the
fuzzers were run by hand).  tests/tools/fuzz_parity.py draws random descriptors and batch sizes for
rtlws_spectra_batch, tests/tools/fuzz_parity_f64.py for rtlws_spectra_batch_f64 -- both against the f64 oracle
(src/spectrum.c:15-99, src/cbb_main.c:106-135 restated); an assertion inside them exits non-zero.  The seeds are
fixed (a gate must be deterministic: the same cases every run, the first ones of sequences the round's soak walked
for 45-60 s each, profiles/r05_fuzz_soak.txt, r05_soak.txt); new cases are the soak scripts' job (tools/r5_soak.sh)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed", [("fuzz_parity.py", 511), ("fuzz_parity_f64.py", 51)])
def test_short_fuzz_seed(built, script, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", script), str(seed), "8"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, "seed %d\n%s\n%s" % (seed, r.stdout[-2000:], r.stderr[-4000:])
    assert "cases ok" in r.stdout and "seed %d" % seed in r.stdout, r.stdout
    cases = int(r.stdout.split("cases ok")[0].split()[-1])
    assert cases >= 5, r.stdout
