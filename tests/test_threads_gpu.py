"""Concurrent callers of the drop-in API: per-thread spectrum handles plus the
shared handle-less entry points (cic_decimate / halfband_decimate share one
device context behind a mutex).  ctypes releases the GIL during the calls."""
import threading

import numpy as np
import pytest

from helpers import rel_err, EPS_K1, TOL

pytestmark = pytest.mark.gpu


def test_four_threads_share_the_engine(built, oracle):
    from rtlws import synth
    errors = []

    def worker(tid):
        try:
            rng = np.random.default_rng(100 + tid)
            s = built.Spectrum(1024)
            iq = synth.tone_noise_iq(8, 1024, seed=tid)
            for rep in range(40):
                k = rep % 8
                ps, ref = np.zeros(1024), np.zeros(1024)
                assert s.add_cmplx_u8(iq[k], ps) == 0
                oracle.spectrum_add_cmplx_u8(1024, iq[k], ref)
                assert rel_err(ps, ref, EPS_K1).max() <= TOL
                R = (8, 10, 12, 3)[tid]
                src = rng.integers(0, 256, size=(R * 257, 2), dtype=np.uint8)
                st0 = rng.integers(-50, 50, size=4).astype(np.int32)
                rc, dst, st = built.cic_decimate(R, src, state=st0)
                rc2, dst2, st2 = oracle.cic_decimate(R, src, state=st0)
                assert rc == rc2 == 0 and np.array_equal(dst, dst2) and np.array_equal(st, st2)
                x = rng.standard_normal(2 * 333).astype(np.float32)
                d1 = rng.standard_normal(10).astype(np.float32)
                d2 = d1.copy()
                assert np.array_equal(built.halfband_decimate(x, d1), oracle.halfband_decimate(x, d2))
                assert np.array_equal(d1, d2)
            s.free()
        except Exception as e:           # surface failures from worker threads
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
