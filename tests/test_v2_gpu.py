"""spectrum_fused_v2.hip (two virtual threads per lane, 4096-point cmplx_u8 frames) against
the four-wavefront kernel it replaces: the arithmetic is the same instruction for
instruction, so every output must be BIT-identical, for every window / K / output mode --
and both are held to the oracle by the rest of the suite (which now runs the v2 kernel
wherever it applies)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(engine, iq, n_fft=4096, **kw):
    """The kernel choice is an engine option (rtlws_engine_set_option), frozen between calls:
    the environment is only read when an engine is created."""
    dflt = engine.get_option("v2")
    with engine.option("v2", 1):
        a = engine.spectra(iq, n_fft, **kw)
    with engine.option("v2", 0):
        b = engine.spectra(iq, n_fft, **kw)
    assert engine.get_option("v2") == dflt       # the engine's own rule is back
    return a, b


@pytest.mark.parametrize("window", ["rect", "hann"])
@pytest.mark.parametrize("k_avg,output", [(1, "power_sum"), (8, "mean_db"), (6, "payload_u8"), (3, "power_sum")])
def test_v2_bit_identical_to_the_four_wavefront_kernel(engine, built, window, k_avg, output):
    from rtlws import synth
    rows = 1031                                   # more rows than resident workgroups: the persistent loop strides
    iq = synth.tone_noise_iq(rows * k_avg, 4096, seed=11 + k_avg)
    iq[5] = 128                                   # a constant frame (all bins zero)
    iq[7] = synth.uniform_iq(1, 4096, seed=3)[0]
    desc = built.make_desc(4096, k_avg, "cu8", window, output)
    with engine.option("v2", 1):
        rc, blocks, threads, lds = engine.grid(desc, rows * k_avg)
    assert (rc, threads) == (0, 128) and lds == 8 * (15 * 290 + 15 * 18 + 18) and blocks <= 4 * 256
    a, b = _both(engine, iq, k_avg=k_avg, window=window, output=output)
    assert a.shape == (rows, 4096)
    assert np.array_equal(a, b, equal_nan=True)


def test_v2_small_and_ragged_grids(engine, oracle):
    from rtlws import synth
    from helpers import rel_err, EPS_K1, TOL
    for rows in (1, 2, 5):
        iq = synth.tone_noise_iq(rows, 4096, seed=rows)
        a, b = _both(engine, iq)
        assert np.array_equal(a, b)
        assert rel_err(a, oracle.batch_spectra_u8(iq, 4096), EPS_K1).max() <= TOL


@pytest.mark.parametrize("window", ["rect", "hann"])
@pytest.mark.parametrize("k_avg,output", [(1, "power_sum"), (8, "mean_db"), (6, "payload_u8"), (3, "power_sum")])
def test_v2_2048_bit_identical_to_the_two_wavefront_kernel(engine, built, oracle, window, k_avg, output):
    """N = 2048: one wavefront per frame (no barrier), a lane's two virtual threads store four
    consecutive outputs; every output bit-identical to spectrum_fused.hip's."""
    from rtlws import synth
    from helpers import rel_err, EPS_K1, TOL
    rows = 2 * 8 * 256 + 5
    iq = synth.tone_noise_iq(rows * k_avg, 2048, seed=21 + k_avg)
    iq[5] = 128
    iq[7] = synth.uniform_iq(1, 2048, seed=3)[0]
    desc = built.make_desc(2048, k_avg, "cu8", window, output)
    with engine.option("v2", 1):
        rc, blocks, threads, lds = engine.grid(desc, rows * k_avg)
    assert (rc, threads) == (0, 64) and lds == 8 * 2334 and blocks <= 8 * 256
    a, b = _both(engine, iq, 2048, k_avg=k_avg, window=window, output=output)
    assert a.shape == (rows, 2048) and np.array_equal(a, b, equal_nan=True)
    if output == "power_sum" and k_avg == 1:
        w = synth.hann(2048) if window == "hann" else None
        assert rel_err(a[:64], oracle.batch_spectra_u8(iq[:64], 2048, window=w), EPS_K1).max() <= TOL
