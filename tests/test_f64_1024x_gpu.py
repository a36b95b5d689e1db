"""spectrum_f64_1024x.hip -- 1024 = 4 x 16 x 16 with the radix-4 done first on packed int16 samples,
a cross-row v_permlane transpose and ONE LDS transposition -- through rtlws_spectra_batch_f64:
against the f64 oracle under the strict metric (floor 1e-9, bound 1e-10), against the
two-transposition kernel it replaces for rectangular 1024-point cmplx_u8 frames (engine option
f64_x1024 = 0), exact zeros on a constant frame, the DC-slot weights, every output mode it serves."""
import numpy as np
import pytest

from helpers import rel_err, EPS_STRICT

pytestmark = pytest.mark.gpu
STRICT_F64 = 1e-10


def _old(engine, iq, **kw):
    with engine.option("f64_x1024", 0):
        return engine.spectra(iq, 1024, f64=True, **kw)


@pytest.mark.parametrize("k_avg", [1, 2, 6, 8])
@pytest.mark.parametrize("rows_f32", [False, True])
def test_x1024_power_sums(engine, oracle, k_avg, rows_f32):
    from rtlws import synth
    rows = 2051                                  # more rows than resident wavefronts: the loop strides
    iq = synth.tone_noise_iq(rows * k_avg, 1024, seed=51 + k_avg)
    iq[3 * k_avg:4 * k_avg] = 128                # a constant row: every bin exactly zero
    iq[4 * k_avg] = synth.pure_tone_iq(1, 1024, seed=2)[0]
    iq[5 * k_avg] = synth.uniform_iq(1, 1024, seed=2)[0]
    iq[6 * k_avg] = 0                            # the extremes of the sample range
    iq[7 * k_avg] = 255
    iq[8 * k_avg, ::2] = 0                       # +-full scale alternating: the largest radix-4 sums
    iq[8 * k_avg, 1::2] = 255
    assert engine.get_option("f64_x1024") == 1
    got = engine.spectra(iq, 1024, k_avg=k_avg, f64=True, rows_f32=rows_f32)
    ref = oracle.batch_spectra_u8(iq, 1024, K=k_avg, nthreads=8)
    if rows_f32:
        assert got.dtype == np.float32 and rel_err(got, ref, EPS_STRICT).max() <= 2.0 ** -24 * 1.001
    else:
        assert got.dtype == np.float64 and rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64
        old = _old(engine, iq[:64 * k_avg], k_avg=k_avg)
        assert rel_err(got[:64], old, EPS_STRICT).max() <= STRICT_F64
    assert not got[3].any()                                           # exact zeros, not 1e-20
    if k_avg == 1:
        assert np.array_equal(got[:, 512], got[:, 511])               # DC-slot rule, K = 1
        assert not got[6].any() and not got[7].any()                  # constant frames again


def test_x1024_dc_weights_and_few_rows(engine, oracle):
    from rtlws import synth
    for rows, k in ((1, 1), (3, 2), (1, 6), (2, 13)):
        iq = synth.uniform_iq(rows * k, 1024, seed=rows + k)
        got = engine.spectra(iq, 1024, k_avg=k, f64=True)
        ref = oracle.batch_spectra_u8(iq, 1024, K=k)
        assert rel_err(got, ref, EPS_STRICT).max() <= STRICT_F64
        # slot N/2 = sum_k (K - k) P_k[N-1] (src/spectrum.c:25-33), not K times its neighbour
        per = oracle.batch_spectra_u8(iq, 1024, K=1).reshape(rows, k, 1024)
        want = sum((k - j) * per[:, j, 511] for j in range(k))
        assert np.allclose(got[:, 512], want, rtol=1e-12)


def test_x1024_db_and_payload_single_frames(engine, oracle):
    from rtlws import synth
    iq = synth.tone_noise_iq(700, 1024, seed=8)
    ref = oracle.batch_spectra_u8(iq, 1024)
    db = engine.spectra(iq, 1024, output="mean_db", f64=True)
    assert np.abs(db - 10 * np.log10(ref)).max() <= 1e-9
    db32 = engine.spectra(iq, 1024, output="mean_db", f64=True, rows_f32=True)
    assert np.array_equal(db32, db.astype(np.float32))
    for gain in (0, 15, -25):
        got = engine.spectra(iq, 1024, output="payload_u8", gain_db=gain, f64=True)
        want = np.stack([oracle.spectrum_payload(r, 1, gain) for r in ref])
        assert got.dtype == np.uint8 and np.array_equal(got, want)
    # K > 1 with a dB / payload epilogue is served by the two-transposition kernel: same contract
    iq6 = synth.tone_noise_iq(6 * 50, 1024, seed=9)
    ref6 = oracle.batch_spectra_u8(iq6, 1024, K=6)
    got6 = engine.spectra(iq6, 1024, k_avg=6, output="payload_u8", f64=True)
    assert np.array_equal(got6, np.stack([oracle.spectrum_payload(r, 6, 0) for r in ref6]))


def test_x1024_full_size_config2(engine, oracle):
    """BASELINE.json configs[1] at full size through the one-transposition kernel, f64 rows."""
    from rtlws import synth
    iq = synth.tone_noise_iq(65536, 1024, seed=4321)
    got = engine.spectra(iq, 1024, f64=True)
    ref = oracle.batch_spectra_u8(iq, 1024, nthreads=16)
    worst = max(float(rel_err(got[lo:lo + 8192], ref[lo:lo + 8192], EPS_STRICT).max()) for lo in range(0, 65536, 8192))
    assert worst <= STRICT_F64, worst


@pytest.mark.parametrize("rows,k_avg", [(1, 1), (7, 1), (255, 1), (257, 3), (2049, 1), (9001, 2)])
@pytest.mark.parametrize("rows_f32", [False, True])
def test_x1024_eight_wavefront_workgroups(engine, oracle, rows, k_avg, rows_f32):
    """The WAVES = 8 form (one workgroup per CU, its wavefronts take the workgroup's rows one at a time from an
    LDS counter; batches of >= 32 rows per CU take it by themselves) forced on batches of every shape -- fewer rows
    than workgroups, than wavefronts, ragged -- against the oracle and, bit for bit, against the one-wavefront
    workgroups: only the row-to-wavefront map differs (src/spectrum.c:15-35,47-63, K loop of src/cbb_main.c:50-59)."""
    from rtlws import synth
    iq = synth.tone_noise_iq(rows * k_avg, 1024, seed=900 + rows)
    iq[0] = 128
    with engine.option("f64_x_waves", 8):
        got = engine.spectra(iq, 1024, k_avg=k_avg, f64=True, rows_f32=rows_f32)
    with engine.option("f64_x_waves", 1):
        one = engine.spectra(iq, 1024, k_avg=k_avg, f64=True, rows_f32=rows_f32)
    assert engine.get_option("f64_x_waves") == 0
    assert np.array_equal(got, one)
    ref = oracle.batch_spectra_u8(iq, 1024, K=k_avg, nthreads=8)
    assert rel_err(got, ref, EPS_STRICT).max() <= (2.0 ** -24 * 1.001 if rows_f32 else STRICT_F64)


def test_x1024_eight_wavefront_db_and_payload_rows(engine, oracle):
    from rtlws import synth
    iq = synth.tone_noise_iq(3000, 1024, seed=41)
    ref = oracle.batch_spectra_u8(iq, 1024, nthreads=8)
    with engine.option("f64_x_waves", 8):
        db = engine.spectra(iq, 1024, f64=True, output="mean_db")
        pay = engine.spectra(iq, 1024, f64=True, output="payload_u8", gain_db=15)
    ok = ref > 1e-9 * ref.max(axis=1, keepdims=True)
    assert np.abs(db - 10 * np.log10(ref))[ok].max() <= 1e-9
    assert np.array_equal(pay, np.stack([oracle.spectrum_payload(r, 1, 15) for r in ref]))
