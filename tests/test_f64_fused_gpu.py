"""spectrum_f64_fused.hip -- rtlws_spectra_batch_f64 on 1024-point cmplx_u8 frames at batch
rates -- against the f64 oracle under the STRICT metric (floor 1e-9 of the row maximum,
bound 1e-10: north_star's 1e-4 with six orders to spare), against the row-per-workgroup
f64 kernel it replaces for that shape, and byte for byte on the payload."""
import numpy as np
import pytest

from helpers import rel_err, EPS_STRICT

pytestmark = pytest.mark.gpu
STRICT_F64 = 1e-10


def _old_kernel(engine, iq, n_fft=1024, **kw):
    with engine.option("f64_fused", 0):       # the row-per-workgroup kernel (spectrum_f64.hip)
        return engine.spectra(iq, n_fft, f64=True, **kw)


@pytest.mark.parametrize("window", ["rect", "hann"])
@pytest.mark.parametrize("k_avg", [1, 6, 8])
def test_f64_fused_vs_oracle_and_old_kernel(engine, oracle, window, k_avg):
    from rtlws import synth
    rows = 2051                                  # more rows than resident wavefronts (2 048): the loop strides
    iq = synth.tone_noise_iq(rows * k_avg, 1024, seed=31 + k_avg)
    iq[3] = 128                                  # constant frame: every bin exactly zero
    iq[4] = synth.pure_tone_iq(1, 1024, seed=2)[0]
    iq[5] = synth.uniform_iq(1, 1024, seed=2)[0]
    got = engine.spectra(iq, 1024, k_avg=k_avg, window=window, f64=True)
    assert got.dtype == np.float64 and got.shape == (rows, 1024)
    w = synth.hann(1024) if window == "hann" else None
    ref = oracle.batch_spectra_u8(iq, 1024, K=k_avg, window=w, nthreads=8)
    assert rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64
    old = _old_kernel(engine, iq[:64 * k_avg], k_avg=k_avg, window=window)
    assert rel_err(got[:64], old, EPS_STRICT).max() <= STRICT_F64
    if k_avg == 1:
        assert np.array_equal(got[:, 512], got[:, 511])          # DC-slot rule, K = 1
        assert not got[3].any() if window == "rect" else True


def test_f64_fused_db_and_payload(engine, oracle):
    from rtlws import synth
    iq = synth.tone_noise_iq(6 * 300, 1024, seed=77)
    ref = oracle.batch_spectra_u8(iq, 1024, K=6, nthreads=8)
    db = engine.spectra(iq, 1024, k_avg=6, output="mean_db", f64=True)
    assert np.abs(db - 10 * np.log10(ref / 6)).max() <= 1e-9
    for gain in (0, 15, -25):
        got = engine.spectra(iq, 1024, k_avg=6, output="payload_u8", gain_db=gain, f64=True)
        want = np.stack([oracle.spectrum_payload(r, 6, gain) for r in ref])
        assert got.dtype == np.uint8 and np.array_equal(got, want)      # identical bytes, no +-1 allowance


def test_f64_fused_few_rows_and_dc_weights(engine, oracle):
    from rtlws import synth
    for rows, k in ((1, 1), (3, 2), (1, 6)):
        iq = synth.uniform_iq(rows * k, 1024, seed=rows + k)
        got = engine.spectra(iq, 1024, k_avg=k, f64=True)
        ref = oracle.batch_spectra_u8(iq, 1024, K=k)
        assert rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64


@pytest.mark.parametrize("N", [2048, 4096])
@pytest.mark.parametrize("window,k_avg,output", [("rect", 1, "power_sum"), ("hann", 8, "mean_db"),
                                                  ("hann", 1, "power_sum"), ("rect", 6, "payload_u8"),
                                                  ("hann", 3, "payload_u8")])
def test_f64_fused_2048_and_4096(engine, oracle, N, window, k_avg, output):
    """The multi-wavefront sizes (two / four wavefronts per frame, barriers, LDS DC slot, direct
    stores): strict metric against the oracle, payload bytes identical, and agreement with the
    row-per-workgroup kernel."""
    from rtlws import synth
    rows = 2 * 256 * (8 // (N // 1024)) // 8 + 3          # more rows than resident workgroups, ragged
    iq = synth.tone_noise_iq(rows * k_avg, N, seed=N + k_avg)
    iq[1] = 128
    iq[2] = synth.pure_tone_iq(1, N, seed=5)[0]
    got = engine.spectra(iq, N, k_avg=k_avg, window=window, output=output, gain_db=15, f64=True)
    w = synth.hann(N) if window == "hann" else None
    ref = oracle.batch_spectra_u8(iq, N, K=k_avg, window=w, nthreads=8)
    if output == "payload_u8":
        want = np.stack([oracle.spectrum_payload(r, k_avg, 15) for r in ref])
        assert got.dtype == np.uint8 and np.array_equal(got, want)
    elif output == "mean_db":
        ok = ref > 0
        assert np.abs(got[ok] - 10 * np.log10(ref[ok] / k_avg)).max() <= 1e-9
    else:
        assert got.dtype == np.float64 and rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64
        old = _old_kernel(engine, iq[:16 * k_avg], N, k_avg=k_avg, window=window)
        assert rel_err(got[:16], old, EPS_STRICT).max() <= STRICT_F64
        if k_avg == 1:
            assert np.array_equal(got[:, N // 2], got[:, N // 2 - 1])


def test_f64_fused_config3_shape(engine, oracle):
    """BASELINE.json configs[2] in the reference's precision: 4096-point Hann, K = 8, mean dB."""
    from rtlws import synth
    iq = synth.tone_noise_iq(8 * 40, 4096, seed=3)
    got = engine.spectra(iq, 4096, k_avg=8, window="hann", output="mean_db", f64=True)
    ref = oracle.batch_spectra_u8(iq, 4096, K=8, window=synth.hann(4096), nthreads=8)
    assert np.abs(got - 10 * np.log10(ref / 8)).max() <= 1e-9
