"""The oracle's DFT against a SECOND implementation of the FFTW3 API, on the CPU.

tests/test_fftw_api_gpu.py makes the reference's call sequence -- fftw_plan_dft_1d(N, in, out, FFTW_FORWARD,
FFTW_ESTIMATE) + fftw_execute on interleaved complex doubles, src/spectrum.c:21,42 -- against ROCm's hipFFTW,
which needs a GPU.  This image also carries Intel MKL's runtime (/opt/conda/lib/libmkl_rt.so: torch's BLAS),
whose FFTW3 interface exports the same entry points; here they are called DIRECTLY through ctypes (no header is
written, nothing of the reference is compiled: FFTW3 itself is absent and src/spectrum.c stays unbuildable,
DESIGN.md 2) with the same arguments the reference passes, and the oracle's DFT -- the thing every GPU parity
test is checked against -- must return the same numbers: sign of the exponent, no normalisation, interleaved
re/im layout.  Then src/spectrum.c:54-60,23-34 around that transform (conversion, shift, DC-slot rule,
K-frame accumulation), written out literally, against the oracle's spectrum_add_cmplx_u8 / _s32 / _real_f32.
Skipped where the library is absent (the GPU box may not carry it).  It still does not execute spectrum.c:
parity of the spectrum stage remains "unpinned by reference execution"; this is one more independent witness
of the call site's conventions, now inside the CPU suite."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

FFTW_FORWARD, FFTW_ESTIMATE = -1, 1 << 6          # fftw3.h's values; src/spectrum.c:42 passes these names


def _find_mkl():
    for pat in ("/opt/conda/lib/libmkl_rt.so*", "/usr/lib/x86_64-linux-gnu/libmkl_rt.so*"):
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


@pytest.fixture(scope="module")
def fftw_api():
    path = _find_mkl()
    if not path:
        pytest.skip("no libmkl_rt.so in this image")
    try:
        L = C.CDLL(path)
        L.fftw_plan_dft_1d.restype = C.c_void_p
    except (OSError, AttributeError) as ex:
        pytest.skip("MKL's FFTW3 interface is not usable here: %s" % ex)
    L.fftw_plan_dft_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    L.fftw_execute.argtypes = [C.c_void_p]
    L.fftw_destroy_plan.argtypes = [C.c_void_p]
    L.fftw_malloc.restype = C.c_void_p
    L.fftw_malloc.argtypes = [C.c_size_t]
    L.fftw_free.argtypes = [C.c_void_p]
    return L


def _fftw_forward(L, x):
    """src/spectrum.c:40-42,21,103-105: fftw_malloc x 2, plan, execute, destroy, fftw_free x 2."""
    n = int(np.asarray(x).size)
    a, b = L.fftw_malloc(16 * n), L.fftw_malloc(16 * n)
    assert a and b
    try:
        vin = np.ctypeslib.as_array((C.c_double * (2 * n)).from_address(a))
        vout = np.ctypeslib.as_array((C.c_double * (2 * n)).from_address(b))
        plan = L.fftw_plan_dft_1d(n, a, b, FFTW_FORWARD, FFTW_ESTIMATE)
        assert plan
        xc = np.ascontiguousarray(x, dtype=np.complex128)
        vin[0::2], vin[1::2] = xc.real, xc.imag          # the reference fills `in` after planning, too (:54-58)
        L.fftw_execute(plan)
        y = vout[0::2] + 1j * vout[1::2]
        L.fftw_destroy_plan(plan)
        return y.copy()
    finally:
        L.fftw_free(a)
        L.fftw_free(b)


@pytest.mark.parametrize("N", [2, 6, 100, 1000, 1024, 2048, 4096, 8192])
def test_oracle_dft_equals_mkls_fftw_interface(fftw_api, oracle, N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    want = _fftw_forward(fftw_api, x)
    got = oracle.dft(x)
    assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max()) * np.log2(max(N, 2))
    e = np.zeros(N, dtype=np.complex128)            # an impulse at n = 1 shows the sign convention by itself
    e[1] = 1.0
    k = np.arange(N)
    assert np.abs(_fftw_forward(fftw_api, e) - np.exp(-2j * np.pi * k / N)).max() <= 1e-14


def _accumulate(ps, X):
    """src/spectrum.c:23-34, literally"""
    N = ps.size
    for i in range(N):
        idx = (N // 2 + i) % N
        if idx > 0:
            ps[i] += X[idx].real ** 2 + X[idx].imag ** 2
        else:
            ps[i] += ps[i - 1]


@pytest.mark.parametrize("N,K", [(1024, 1), (1024, 6), (4096, 8), (100, 3)])
def test_spectrum_stage_over_mkls_fftw_interface_equals_the_oracle(fftw_api, oracle, N, K):
    from rtlws import synth
    iq = synth.tone_noise_iq(K, N, seed=12 + N + K)
    ps, ref = np.zeros(N), np.zeros(N)
    for f in range(K):
        x = (iq[f, :, 0].astype(np.float64) - 128.0) / 128.0 + 1j * (iq[f, :, 1].astype(np.float64) - 128.0) / 128.0   # :56-57
        _accumulate(ps, _fftw_forward(fftw_api, x))
        assert oracle.spectrum_add_cmplx_u8(N, iq[f], ref) == 0
    assert np.abs(ps - ref).max() <= 1e-12 * ref.max()
    # the other two entry points (src/spectrum.c:72-76: s32 / 128; :90-94: (x, 0))
    rng = np.random.default_rng(5)
    s32 = rng.integers(-1024, 1024, size=(N, 2), dtype=np.int32)
    ps, ref = np.zeros(N), np.zeros(N)
    _accumulate(ps, _fftw_forward(fftw_api, s32[:, 0] / 128.0 + 1j * s32[:, 1] / 128.0))
    assert oracle.spectrum_add_cmplx_s32(N, s32, ref) == 0
    assert np.abs(ps - ref).max() <= 1e-12 * ref.max()
    f32 = rng.standard_normal(N).astype(np.float32)
    ps, ref = np.zeros(N), np.zeros(N)
    _accumulate(ps, _fftw_forward(fftw_api, f32.astype(np.float64) + 0j))
    assert oracle.spectrum_add_real_f32(N, f32, ref) == 0
    assert np.abs(ps - ref).max() <= 1e-12 * ref.max()
