"""Round 4 extensions of spectrum_f64_fused.hip, through the C-ABI (rtlws_spectra_batch_f64):

* RTLWS_FLAG_ROWS_F32 -- f64 arithmetic, rows rounded once to f32: the strict metric (floor 1e-9
  of the row maximum) must stay at ONE f32 rounding (<= 6e-8) on every bin of
  every row, whatever the dynamic range -- north_star's 1e-4 with three orders to spare and at
  the contract's own byte count -- and must equal the f64 rows rounded on the host, bit for bit;
* the fused kernel on cmplx_s32 / real-f32 frames (src/spectrum.c:65-99) and behind the CIC-fused
  input stage for R = 8 / 10 / 12 (src/resample.c:21-40 -> src/spectrum.c:65-81): strict <= 1e-10
  against the f64 oracle, agreement with the row-per-workgroup kernel it replaces there, payload
  bytes identical.
"""
import numpy as np
import pytest

from helpers import rel_err, EPS_STRICT

pytestmark = pytest.mark.gpu
STRICT_F64 = 1e-10
ONE_F32_ROUNDING = 2.0 ** -24 * 1.001         # half an ulp, relative


def _general(engine, data, n_fft, **kw):
    with engine.option("f64_fused", 0):           # spectrum_f64.hip, one workgroup per row
        return engine.spectra(data, n_fft, f64=True, **kw)


# ---- f64 arithmetic, f32 rows ------------------------------------------------------------

@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("window,k_avg,output", [("rect", 1, "power_sum"), ("hann", 1, "power_sum"),
                                                  ("rect", 6, "power_sum"), ("hann", 8, "mean_db")])
def test_rows_f32_are_the_f64_rows_rounded_once(engine, oracle, N, window, k_avg, output):
    from rtlws import synth
    rows = 2 * 256 * (8 // (N // 1024)) // 8 + 3          # more rows than resident workgroups, ragged
    iq = synth.tone_noise_iq(rows * k_avg, N, seed=N + 7 * k_avg)
    iq[1] = 128                                           # constant frame: every bin zero (-inf dB)
    iq[2] = synth.pure_tone_iq(1, N, seed=5)[0]           # the worst dynamic range a u8 frame has
    got32 = engine.spectra(iq, N, k_avg=k_avg, window=window, output=output, f64=True, rows_f32=True)
    got64 = engine.spectra(iq, N, k_avg=k_avg, window=window, output=output, f64=True)
    assert got32.dtype == np.float32 and got32.shape == (rows, N) and got64.dtype == np.float64
    with np.errstate(over="ignore"):
        assert np.array_equal(got32, got64.astype(np.float32), equal_nan=True)
    w = synth.hann(N) if window == "hann" else None
    ref = oracle.batch_spectra_u8(iq, N, K=k_avg, window=w, nthreads=8)
    if output == "mean_db":
        ok = ref > 0
        assert np.abs(got32[ok] - 10 * np.log10(ref[ok] / k_avg)).max() <= 1e-5     # f32 ulp at ~100 dB
    else:
        # strict metric, no builder-chosen floor: one f32 rounding everywhere
        assert rel_err(got32, ref, EPS_STRICT).max() <= ONE_F32_ROUNDING
        if k_avg == 1:
            assert np.array_equal(got32[:, N // 2], got32[:, N // 2 - 1])           # DC-slot rule


def test_rows_f32_general_kernel_and_payload_ignore(engine, oracle):
    """The flag on the row-per-workgroup kernel (a size the fused kernel does not cover), and its
    documented no-op on payload bytes."""
    from rtlws import synth
    iq = synth.tone_noise_iq(12, 512, seed=3)
    got = engine.spectra(iq, 512, k_avg=2, f64=True, rows_f32=True)
    ref = oracle.batch_spectra_u8(iq, 512, K=2)
    assert got.dtype == np.float32 and rel_err(got, ref, EPS_STRICT).max() <= ONE_F32_ROUNDING
    iq = synth.tone_noise_iq(12, 1024, seed=4)
    a = engine.spectra(iq, 1024, k_avg=6, output="payload_u8", gain_db=15, f64=True, rows_f32=True)
    b = engine.spectra(iq, 1024, k_avg=6, output="payload_u8", gain_db=15, f64=True)
    assert a.dtype == np.uint8 and np.array_equal(a, b)


def test_rows_f32_config2_full_size(engine, oracle):
    """BASELINE.json configs[1] at its full size (65 536 frames of 1024 cmplx_u8): every row of the
    f64-arithmetic / f32-row kernel against the f64 oracle under the strict metric."""
    from rtlws import synth
    iq = synth.tone_noise_iq(65536, 1024, seed=1234)
    got = engine.spectra(iq, 1024, f64=True, rows_f32=True)
    ref = oracle.batch_spectra_u8(iq, 1024, nthreads=16)
    worst = 0.0
    for lo in range(0, 65536, 8192):                       # bounded temporaries
        worst = max(worst, float(rel_err(got[lo:lo + 8192], ref[lo:lo + 8192], EPS_STRICT).max()))
    assert worst <= ONE_F32_ROUNDING, worst


# ---- the other input kinds on the fused kernel ---------------------------------------------

@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("window,k_avg", [("rect", 1), ("hann", 3)])
def test_f64_fused_s32_and_f32_inputs(engine, oracle, N, window, k_avg):
    from rtlws import synth
    rng = np.random.default_rng(N + k_avg)
    rows = 2 * 256 * (8 // (N // 1024)) // 8 + 2
    w = synth.hann(N) if window == "hann" else None
    # cmplx_s32: CIC-sized values, plus one frame at the int32 extremes
    s32 = rng.integers(-128 * 12, 128 * 12, size=(rows * k_avg, N, 2), dtype=np.int32)
    s32[k_avg] = rng.integers(-2**31, 2**31 - 1, size=(N, 2), dtype=np.int64).astype(np.int32)
    got = engine.spectra(s32, N, k_avg=k_avg, input="cs32", window=window, f64=True)
    probe = [0, 1, rows // 2, rows - 1]
    for r in probe:
        ps = np.zeros(N)
        for k in range(k_avg):
            assert oracle.spectrum_add_cmplx_s32(N, s32[r * k_avg + k], ps, window=w) == 0
        assert rel_err(got[r], ps, EPS_STRICT).max() <= STRICT_F64, r
    old = _general(engine, s32[:8 * k_avg], N, k_avg=k_avg, input="cs32", window=window)
    assert rel_err(got[:8], old, EPS_STRICT).max() <= STRICT_F64
    # real f32
    f32 = rng.standard_normal(size=(rows * k_avg, N)).astype(np.float32)
    gotf = engine.spectra(f32, N, k_avg=k_avg, input="rf32", window=window, f64=True)
    for r in probe:
        ps = np.zeros(N)
        for k in range(k_avg):
            assert oracle.spectrum_add_real_f32(N, f32[r * k_avg + k], ps, window=w) == 0
        assert rel_err(gotf[r], ps, EPS_STRICT).max() <= STRICT_F64, r
    oldf = _general(engine, f32[:8 * k_avg], N, k_avg=k_avg, input="rf32", window=window)
    assert rel_err(gotf[:8], oldf, EPS_STRICT).max() <= STRICT_F64


@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("R", [8, 10, 12])
def test_f64_fused_cic_input_stage(engine, oracle, N, R):
    """cmplx_u8 -> CIC R:1 -> N-point spectrum in double, one kernel: the integer block sums are
    exact, so the result is held to the same strict bound as the plain u8 kernel."""
    from rtlws import synth
    rows = 2 * 256 * (8 // (N // 1024)) // 8 + 3
    for window, k_avg, output in (("rect", 1, "power_sum"), ("hann", 2, "power_sum"), ("rect", 4, "mean_db"),
                                  ("rect", 3, "payload_u8")):
        nrows = rows if k_avg == 1 else 37
        iq = synth.tone_noise_iq(nrows * k_avg * R, N, seed=R + N + k_avg)      # nrows*k frames of N*R samples
        iq = iq.reshape(nrows * k_avg, N * R, 2)
        iq[1] = 128
        iq[2] = synth.uniform_iq(R, N, seed=9).reshape(N * R, 2)
        w = synth.hann(N) if window == "hann" else None
        ref = oracle.batch_spectra_cic_u8(iq, N, R, K=k_avg, window=w, nthreads=8)
        got = engine.spectra(iq, N, k_avg=k_avg, cic_r=R, window=window, output=output, gain_db=-25, f64=True)
        if output == "payload_u8":
            want = np.stack([oracle.spectrum_payload(r, k_avg, -25) for r in ref])
            assert got.dtype == np.uint8 and np.array_equal(got, want)
        elif output == "mean_db":
            ok = ref > 0
            assert np.abs(got[ok] - 10 * np.log10(ref[ok] / k_avg)).max() <= 1e-9
        else:
            assert rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64
            old = _general(engine, iq[:4 * k_avg], N, k_avg=k_avg, cic_r=R, window=window)
            assert rel_err(got[:4], old, EPS_STRICT).max() <= STRICT_F64
            g32 = engine.spectra(iq, N, k_avg=k_avg, cic_r=R, window=window, f64=True, rows_f32=True)
            assert np.array_equal(g32, got.astype(np.float32))


def test_f64_config4_shape_vs_composed_reference_calls(engine, oracle):
    """BASELINE.json configs[3] in the reference's precision, as the reference would compose it:
    cic_decimate (src/resample.c:6-45) into cmplx_s32, then spectrum_add_cmplx_s32
    (src/spectrum.c:65-81) -- both through the oracle's restatements -- against ONE launch of the
    fused f64 kernel on the raw cmplx_u8 stream."""
    from rtlws import synth
    N, R, rows = 2048, 8, 24
    iq = synth.tone_noise_iq(rows * R, N, seed=44).reshape(rows, N * R, 2)
    got = engine.spectra(iq, N, cic_r=R, f64=True)
    state = None
    for r in range(rows):
        rc, dec, state = oracle.cic_decimate(R, iq[r], state)
        assert rc == 0
        ps = np.zeros(N)
        assert oracle.spectrum_add_cmplx_s32(N, dec, ps) == 0
        assert rel_err(got[r], ps, EPS_STRICT).max() <= STRICT_F64


def test_f64_unaligned_pointers_take_the_general_kernel(engine, built, oracle):
    """d_out 8-byte but not 16-byte aligned: the descriptor is still served (by spectrum_f64.hip)."""
    from rtlws import synth
    iq = synth.tone_noise_iq(5, 1024, seed=12)
    d_in = engine.upload(iq)
    d_out = engine.alloc(5 * 1024 * 8 + 16)
    desc = built.make_desc(1024)
    engine.spectra_batch_f64(desc, d_in, 5, d_out.ptr + 8)
    engine.sync()
    full = engine.download(d_out, np.float64, (5 * 1024 + 2,))
    got = full[1:1 + 5 * 1024].reshape(5, 1024)
    assert rel_err(got, oracle.batch_spectra_u8(iq, 1024), EPS_STRICT).max() <= STRICT_F64
