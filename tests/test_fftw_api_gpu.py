"""The oracle's DFT against an implementation of the FFTW3 API itself.

The reference's spectrum stage is `fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, FFTW_ESTIMATE)` +
`fftw_execute` on interleaved complex doubles (reference src/spectrum.c:21,42).  FFTW3 is not in
this image, so src/spectrum.c cannot be built (parity unpinned by reference
execution) -- but ROCm ships hipFFTW, a library that exports exactly those FFTW3 entry points
(/opt/rocm/lib/libhipfftw.so, f64 through rocFFT; it needs a GPU).  This test makes the reference's
own call sequence against it and requires the oracle's DFT (oracle/rtlws_oracle.c, the thing every
GPU parity test is checked against) to return the same numbers: sign of the exponent,
no normalisation, interleaved re/im layout -- the conventions of the call site, observed through
the API rather than restated from its manual.  It does not execute spectrum.c; the conversion,
shift and DC-slot code stay a line-cited restatement."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FFTW_FORWARD, FFTW_ESTIMATE = -1, 1 << 6          # fftw3.h's values; src/spectrum.c:42 passes these names
HIPFFTW = "/opt/rocm/lib/libhipfftw.so"


@pytest.fixture(scope="module")
def fftw_api():
    if not os.path.exists(HIPFFTW):
        pytest.fail("libhipfftw.so is part of the ROCm image this repo targets")
    L = C.CDLL(HIPFFTW)
    L.fftw_plan_dft_1d.restype = C.c_void_p
    L.fftw_plan_dft_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    L.fftw_execute.argtypes = [C.c_void_p]
    L.fftw_destroy_plan.argtypes = [C.c_void_p]
    return L


def _fftw_forward(L, x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.zeros_like(x)
    plan = L.fftw_plan_dft_1d(x.size, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                              FFTW_FORWARD, FFTW_ESTIMATE)
    assert plan
    L.fftw_execute(plan)
    L.fftw_destroy_plan(plan)
    return y


@pytest.mark.parametrize("N", [2, 6, 100, 1000, 1024, 2048, 4096, 8192])
def test_oracle_dft_equals_the_fftw_api_forward_transform(fftw_api, oracle, N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    want = _fftw_forward(fftw_api, x)
    got = oracle.dft(x)
    assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max()) * np.log2(max(N, 2))
    # an impulse at n = 1 shows the sign convention by itself: exp(-2 pi i k / N)
    e = np.zeros(N, dtype=np.complex128)
    e[1] = 1.0
    k = np.arange(N)
    assert np.abs(_fftw_forward(fftw_api, e) - np.exp(-2j * np.pi * k / N)).max() <= 1e-14
    assert np.abs(oracle.dft(e) - np.exp(-2j * np.pi * k / N)).max() <= 1e-14


def test_spectrum_stage_over_the_fftw_api_equals_the_oracle(fftw_api, oracle):
    """src/spectrum.c:54-60 + :23-34 around that transform -- conversion (u8 - 128) / 128, FFTW_FORWARD,
    |X|^2 into slot (i + N/2) % N, slot N/2 from its already-updated left neighbour, accumulated over
    K frames -- written here in numpy on top of the FFTW-API transform, against the oracle's
    spectrum_add_cmplx_u8 (the accumulation order of the DC slot included)."""
    from rtlws import synth
    N, K = 1024, 6
    iq = synth.tone_noise_iq(K, N, seed=12)
    ps = np.zeros(N)
    for f in range(K):
        x = (iq[f, :, 0].astype(np.float64) - 128.0) / 128.0 + 1j * (iq[f, :, 1].astype(np.float64) - 128.0) / 128.0
        X = _fftw_forward(fftw_api, x)
        for i in range(N):                      # src/spectrum.c:23-34, literally
            idx = (N // 2 + i) % N
            if idx > 0:
                ps[i] += X[idx].real ** 2 + X[idx].imag ** 2
            else:
                ps[i] += ps[i - 1]
    ref = np.zeros(N)
    for f in range(K):
        assert oracle.spectrum_add_cmplx_u8(N, iq[f], ref) == 0
    assert np.abs(ps - ref).max() <= 1e-12 * ref.max()
