"""GPU parity at BASELINE.json's full sizes, through size-independent
properties (every row checked) plus the oracle on a random subsample.

  Parseval:   sum_k |X[k]|^2 = N * sum_n |x[n]|^2.  The output row shows every
              bin except bin 0, whose slot repeats bin N-1 (DC-slot rule), so
              sum(row) - row[N/2] + |X[0]|^2 must equal N * energy, and |X[0]|^2
              is just |sum_n x[n]|^2.
  DC slot:    K = 1 -> row[N/2] == row[N/2 - 1] bit for bit.
"""
import numpy as np
import pytest

from helpers import rel_err, strict_stats, EPS_K1, EPS_STRICT, TOL, STRICT_K1_P999, STRICT_K1_MAX

pytestmark = pytest.mark.gpu


def _energy_and_dc(iq, R=1, window=None, chunk=4096):
    """Per frame: sum |x|^2 and |sum x|^2 of the f64 FFT input (x = (u8-128)/128,
    or CIC block sums / 128 when R > 1)."""
    n = iq.shape[0]
    energy = np.empty(n)
    dc = np.empty(n)
    for a in range(0, n, chunk):
        x = iq[a:a + chunk].astype(np.float64) - 128.0
        if R > 1:
            x = x.reshape(x.shape[0], -1, R, 2).sum(axis=2)
        x /= 128.0
        if window is not None:
            x = x * window[None, :, None]
        energy[a:a + chunk] = (x ** 2).sum(axis=(1, 2))
        s = x.sum(axis=1)
        dc[a:a + chunk] = s[:, 0] ** 2 + s[:, 1] ** 2
    return energy, dc


def test_config2_65536_frames_of_1024(engine, oracle):
    nframes, N = 65536, 1024
    rng = np.random.default_rng(2026)
    iq = rng.integers(0, 256, size=(nframes, N, 2), dtype=np.uint8)
    # a full-scale tone in every 64th frame so strong bins are exercised too
    n = np.arange(N)
    for f in range(0, nframes, 64):
        k = (f // 64) % N
        ph = 2 * np.pi * k * n / N
        iq[f, :, 0] = np.clip(np.round(100 * np.cos(ph) + 128), 0, 255)
        iq[f, :, 1] = np.clip(np.round(100 * np.sin(ph) + 128), 0, 255)
    got = engine.spectra(iq, N)
    assert got.shape == (nframes, N) and np.isfinite(got).all()
    assert np.array_equal(got[:, N // 2], got[:, N // 2 - 1])            # DC-slot rule, K = 1
    energy, dc = _energy_and_dc(iq)
    lhs = got.sum(axis=1, dtype=np.float64) - got[:, N // 2].astype(np.float64) + dc
    assert (np.abs(lhs - N * energy) / (N * energy)).max() < 2e-6         # Parseval, every row
    tones = np.arange(0, nframes, 64)
    peak = got[tones].argmax(axis=1)
    want = ((tones // 64) % N + N // 2) % N                               # fft-shifted bin
    ok = want != N // 2                                                   # bin 0 is never shown
    assert np.array_equal(peak[ok], want[ok])
    rows = rng.choice(nframes, size=768, replace=False)
    ref = oracle.batch_spectra_u8(iq[rows], N, nthreads=8)
    assert rel_err(got[rows], ref, EPS_K1).max() <= TOL
    mx, p999 = strict_stats(got[rows], ref)          # strict floor (1e-9) guard, K = 1
    assert mx <= STRICT_K1_MAX and p999 <= STRICT_K1_P999, (mx, p999)


def test_config3_16384_frames_of_4096_hann_k8(engine, oracle):
    from rtlws import synth
    nframes, N, K = 16384, 4096, 8
    rng = np.random.default_rng(3)
    iq = rng.integers(0, 256, size=(nframes, N, 2), dtype=np.uint8)
    w = synth.hann(N)
    got = engine.spectra(iq, N, k_avg=K, window="hann")                  # power sums
    energy, dc = _energy_and_dc(iq, window=w)
    e_g = energy.reshape(-1, K).sum(axis=1)
    # slot N/2 holds sum_k (K-k) P_k[N-1]; remove it and add the true bin-0 power
    lhs = got.sum(axis=1, dtype=np.float64) - got[:, N // 2].astype(np.float64) + dc.reshape(-1, K).sum(axis=1)
    assert (np.abs(lhs - N * e_g) / (N * e_g)).max() < 2e-6
    rows = rng.choice(nframes // K, size=48, replace=False)
    sel = (rows[:, None] * K + np.arange(K)[None, :]).reshape(-1)
    ref = oracle.batch_spectra_u8(iq[sel], N, K=K, window=w, nthreads=8)
    assert rel_err(got[rows], ref, EPS_STRICT).max() <= TOL
    db = engine.spectra(iq[sel], N, k_avg=K, window="hann", output="mean_db")
    assert np.abs(db - 10 * np.log10(ref / K)).max() <= 2e-4


def test_config4_8192_spectra_cic8_2048(engine, oracle):
    nspec, N, R = 8192, 2048, 8
    rng = np.random.default_rng(4)
    iq = rng.integers(0, 256, size=(nspec, N * R, 2), dtype=np.uint8)
    got = engine.spectra(iq, N, cic_r=R)
    assert np.array_equal(got[:, N // 2], got[:, N // 2 - 1])
    energy, dc = _energy_and_dc(iq, R=R, chunk=512)
    lhs = got.sum(axis=1, dtype=np.float64) - got[:, N // 2].astype(np.float64) + dc
    assert (np.abs(lhs - N * energy) / (N * energy)).max() < 2e-6
    rows = rng.choice(nspec, size=96, replace=False)
    ref = oracle.batch_spectra_cic_u8(iq[rows], N, R, nthreads=8)
    assert rel_err(got[rows], ref, EPS_K1).max() <= TOL
    mx, p999 = strict_stats(got[rows], ref)          # strict floor (1e-9) guard, K = 1
    assert mx <= STRICT_K1_MAX and p999 <= STRICT_K1_P999, (mx, p999)
    # the CIC on its own, bit-exact on the whole 256 MiB input
    d_src = engine.upload(iq)
    d_dst = engine.alloc(nspec * N * 8)
    engine.cic_block_sums(R, d_src, nspec * N, d_dst)
    dec = engine.download(d_dst, np.int32, (nspec * N, 2))
    for a in range(0, nspec, 1024):
        want = (iq[a:a + 1024].astype(np.int32) - 128).reshape(-1, R, 2).sum(axis=1)
        assert np.array_equal(dec[a * N:(a + 1024) * N], want)
