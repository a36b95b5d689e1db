"""ctypes/numpy front end of the CPU oracle (oracle/librtlws_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product path never imports this module.
Reference citations live in rtlws_oracle.h / rtlws_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTLWS_ORACLE_LIB: an alternative build of the same source (e.g. with sanitizers, tests/tools/asan_cpu.sh)
_LIB_PATH = os.environ.get("RTLWS_ORACLE_LIB") or os.path.join(_HERE, "librtlws_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "librtlws_ref.so")


def build(quiet=True):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i, l = C.c_void_p, C.c_int, C.c_long
        L.orc_dft_forward.argtypes = [i, vp, vp]
        L.orc_dft_direct.argtypes = [i, vp, vp]
        for name in ("orc_spectrum_add_cmplx_u8", "orc_spectrum_add_cmplx_s32",
                     "orc_spectrum_add_real_f32"):
            f = getattr(L, name)
            f.argtypes = [i, vp, vp, vp, i]
            f.restype = i
        L.orc_batch_spectra_u8.argtypes = [i, i, l, vp, vp, vp, i]
        L.orc_batch_spectra_u8.restype = i
        L.orc_batch_spectra_cic_u8.argtypes = [i, i, i, l, vp, vp, vp, i]
        L.orc_batch_spectra_cic_u8.restype = i
        L.orc_cic_decimate.argtypes = [i, vp, i, vp, i, vp]
        L.orc_cic_decimate.restype = i
        L.orc_halfband_decimate.argtypes = [vp, vp, i, vp]
        L.orc_rfdec_new.restype = vp
        L.orc_rfdec_set_parameters.argtypes = [vp, C.c_double, i]
        L.orc_rfdec_set_parameters.restype = i
        L.orc_rfdec_decimate.argtypes = [vp, vp, i, vp, vp]
        L.orc_rfdec_decimate.restype = i
        L.orc_rfdec_input_len.argtypes = [vp]
        L.orc_rfdec_input_len.restype = i
        L.orc_rfdec_resampled_len.argtypes = [vp]
        L.orc_rfdec_resampled_len.restype = i
        L.orc_rfdec_free.argtypes = [vp]
        L.orc_atan2_approx.argtypes = [C.c_float, C.c_float]
        L.orc_atan2_approx.restype = C.c_float
        L.orc_fm_demod.argtypes = [vp, i, vp, vp]
        L.orc_audio_block.argtypes = [vp, i, vp, vp]
        L.orc_spectrum_payload.argtypes = [i, vp, i, i, vp]
        L.orc_spectrum_payload.restype = i
        L.orc_estimate_spectrum.argtypes = [vp, i, vp]
        L.orc_estimate_spectrum.restype = i
        L.orc_mean_db.argtypes = [i, vp, i, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _win(window, N):
    if window is None:
        return None
    w = np.ascontiguousarray(window, dtype=np.float64)
    assert w.shape == (N,)
    return w


# ---- DFT -----------------------------------------------------------------

def dft(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_dft_forward(x.size, _p(x), _p(out))
    return out


def dft_direct(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_dft_direct(x.size, _p(x), _p(out))
    return out


# ---- spectrum.c ----------------------------------------------------------

def spectrum_add_cmplx_u8(N, src, ps, window=None, length=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    assert ps.dtype == np.float64 and ps.flags.c_contiguous
    w = _win(window, N)
    return lib().orc_spectrum_add_cmplx_u8(N, _p(src), _p(w), _p(ps),
                                           N if length is None else length)


def spectrum_add_cmplx_s32(N, src, ps, window=None, length=None):
    src = np.ascontiguousarray(src, dtype=np.int32)
    w = _win(window, N)
    return lib().orc_spectrum_add_cmplx_s32(N, _p(src), _p(w), _p(ps),
                                            N if length is None else length)


def spectrum_add_real_f32(N, src, ps, window=None, length=None):
    src = np.ascontiguousarray(src, dtype=np.float32)
    w = _win(window, N)
    return lib().orc_spectrum_add_real_f32(N, _p(src), _p(w), _p(ps),
                                           N if length is None else length)


def batch_spectra_u8(src, N, K=1, window=None, nthreads=1, out=None):
    """src: uint8 [nframes*N*2] (any shape); returns f64 [nframes/K, N].
    out: optional preallocated result (timing loops reuse it: a fresh 128 MiB
    array per call is page-faulted in, which costs more than the transforms)."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1)
    nframes = src.size // (2 * N)
    assert nframes * 2 * N == src.size and nframes % K == 0
    if out is None:
        out = np.empty((nframes // K, N), dtype=np.float64)
    assert out.shape == (nframes // K, N) and out.dtype == np.float64 and out.flags.c_contiguous
    w = _win(window, N)
    rc = lib().orc_batch_spectra_u8(N, K, nframes, _p(src), _p(w), _p(out), nthreads)
    assert rc == 0
    return out


def batch_spectra_cic_u8(src, N, R, K=1, window=None, nthreads=1, out=None):
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1)
    nframes = src.size // (2 * N * R)
    assert nframes * 2 * N * R == src.size and nframes % K == 0
    if out is None:
        out = np.empty((nframes // K, N), dtype=np.float64)
    assert out.shape == (nframes // K, N) and out.dtype == np.float64 and out.flags.c_contiguous
    w = _win(window, N)
    rc = lib().orc_batch_spectra_cic_u8(N, K, R, nframes, _p(src), _p(w), _p(out), nthreads)
    assert rc == 0
    return out


# ---- resample.c ----------------------------------------------------------

def cic_decimate(R, src, state=None, dst_len=None):
    """Returns (rc, dst int32 [dst_len,2], state int32[4])."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1, 2)
    src_len = src.shape[0]
    if dst_len is None:
        dst_len = src_len // R if R > 0 else 0
    dst = np.zeros((max(dst_len, 0), 2), dtype=np.int32)
    st = np.zeros(4, dtype=np.int32) if state is None else np.array(state, dtype=np.int32)
    rc = lib().orc_cic_decimate(R, _p(src), src_len, _p(dst), dst_len, _p(st))
    return rc, dst, st


def halfband_decimate(inp, delay):
    """inp f32 [2*out_len]; delay f32[10] is updated in place. Returns out."""
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    assert delay.dtype == np.float32 and delay.size == 10
    out_len = inp.size // 2
    out = np.empty(out_len, dtype=np.float32)
    lib().orc_halfband_decimate(_p(inp), _p(out), out_len, _p(delay))
    return out


def atan2_approx(y, x):
    return lib().orc_atan2_approx(float(y), float(x))


def fm_demod(iq, prev_phase=0.0):
    """iq int32 [len,2]; returns (demod f32[len], new prev_phase)."""
    iq = np.ascontiguousarray(iq, dtype=np.int32).reshape(-1, 2)
    out = np.empty(iq.shape[0], dtype=np.float32)
    prev = np.array([prev_phase], dtype=np.float32)
    lib().orc_fm_demod(_p(iq), iq.shape[0], _p(prev), _p(out))
    return out, float(prev[0])


def audio_block(iq, state):
    """One decimator block through demod + 2 half-bands; state f32[21] updated in place."""
    iq = np.ascontiguousarray(iq, dtype=np.int32).reshape(-1, 2)
    assert state.dtype == np.float32 and state.size == 21
    out = np.empty(iq.shape[0] // 4, dtype=np.float32)
    lib().orc_audio_block(_p(iq), iq.shape[0], _p(state), _p(out))
    return out


_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


class RfDecimator:
    """src/rf_decimator.c restated; callbacks collected as int32 arrays."""

    def __init__(self):
        self._h = lib().orc_rfdec_new()
        self.blocks = []

    def set_parameters(self, sample_rate, down_factor):
        return lib().orc_rfdec_set_parameters(self._h, float(sample_rate), int(down_factor))

    @property
    def input_len(self):
        return lib().orc_rfdec_input_len(self._h)

    @property
    def resampled_len(self):
        return lib().orc_rfdec_resampled_len(self._h)

    def decimate(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1, 2)

        def _cb(ptr, n, _user):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 2))
            self.blocks.append(a.copy())

        cb = _CB(_cb)
        return lib().orc_rfdec_decimate(self._h, _p(iq), iq.shape[0], cb, None)

    def close(self):
        if self._h:
            lib().orc_rfdec_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


# ---- cbb_main.c ----------------------------------------------------------

def spectrum_payload(ps, count, gain_db):
    ps = np.ascontiguousarray(ps, dtype=np.float64)
    buf = np.zeros(ps.size, dtype=np.uint8)
    n = lib().orc_spectrum_payload(ps.size, _p(ps), int(count), int(gain_db), _p(buf))
    return buf[:n].copy()


def estimate_spectrum(iq):
    iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1, 2)
    ps = np.zeros(1024, dtype=np.float64)
    blocks = lib().orc_estimate_spectrum(_p(iq), iq.shape[0], _p(ps))
    return ps, blocks


def mean_db(ps, count):
    ps = np.ascontiguousarray(ps, dtype=np.float64)
    out = np.empty_like(ps)
    lib().orc_mean_db(ps.size, _p(ps), int(count), _p(out))
    return out


# ---- reference object code (oracle/_ref) -----------------------------------

class _CicDelay(C.Structure):
    # src/resample.h:8-12: two cmplx_s32 (8 bytes each)
    _fields_ = [("int_re", C.c_int32), ("int_im", C.c_int32),
                ("comb_re", C.c_int32), ("comb_im", C.c_int32)]


_ref = None


def ref_available():
    return os.path.exists(_REF_PATH)


def ref():
    """The reference's own resample.c / rf_decimator.c object code, if built."""
    global _ref
    if _ref is None:
        R = C.CDLL(_REF_PATH)
        vp, i = C.c_void_p, C.c_int
        R.cic_decimate.argtypes = [i, vp, i, vp, i, C.POINTER(_CicDelay)]
        R.cic_decimate.restype = i
        R.halfband_decimate.argtypes = [vp, vp, i, vp]
        R.halfband_decimate.restype = None
        R.rf_decimator_alloc.restype = vp
        R.rf_decimator_add_callback.argtypes = [vp, vp]
        R.rf_decimator_set_parameters.argtypes = [vp, C.c_double, i]
        R.rf_decimator_set_parameters.restype = i
        R.rf_decimator_decimate_cmplx_u8.argtypes = [vp, vp, i]
        R.rf_decimator_decimate_cmplx_u8.restype = i
        R.rf_decimator_free.argtypes = [vp]
        R.audio_fm_demodulator.argtypes = [vp, i]
        R.audio_get_audio_payload.argtypes = [vp, i]
        R.audio_get_audio_payload.restype = i
        R.audio_new_audio_available.restype = i
        _ref = R
    return _ref


def ref_cic_decimate(R, src, state=None, dst_len=None):
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1, 2)
    src_len = src.shape[0]
    if dst_len is None:
        dst_len = src_len // R
    dst = np.zeros((dst_len, 2), dtype=np.int32)
    d = _CicDelay(*([0, 0, 0, 0] if state is None else [int(x) for x in state]))
    rc = ref().cic_decimate(R, _p(src), src_len, _p(dst), dst_len, C.byref(d))
    return rc, dst, np.array([d.int_re, d.int_im, d.comb_re, d.comb_im], dtype=np.int32)


def ref_halfband_decimate(inp, delay):
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    out = np.empty(inp.size // 2, dtype=np.float32)
    ref().halfband_decimate(_p(inp), _p(out), out.size, _p(delay))
    return out


_REF_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


def ref_rf_decimate(sample_rate, down_factor, chunks):
    """Feed chunks (list of uint8 [n,2]) through the reference rf_decimator;
    returns (return codes, list of int32 [resampled_len,2] callback blocks)."""
    R = ref()
    blocks = []

    def _cb(ptr, n):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 2))
        blocks.append(a.copy())

    cb = _REF_CB(_cb)
    h = R.rf_decimator_alloc()
    R.rf_decimator_add_callback(h, C.cast(cb, C.c_void_p))
    rcs = [R.rf_decimator_set_parameters(h, float(sample_rate), int(down_factor))]
    for ch in chunks:
        ch = np.ascontiguousarray(ch, dtype=np.uint8).reshape(-1, 2)
        rcs.append(R.rf_decimator_decimate_cmplx_u8(h, _p(ch), ch.shape[0]))
    R.rf_decimator_free(h)
    return rcs, blocks


def ref_audio_chain(blocks):
    """Feed int32 [len,2] blocks (all the same length) through the reference's
    audio_fm_demodulator (src/audio_main.c, object code) in THIS process and
    fetch the audio it queues, one buffer per read.  The reference keeps its
    delay lines in function statics, so call this once per process
    (oracle/gen_golden.py does, in a child process).

    Two quirks of the reference's payload function (src/audio_main.c:40-72), which
    is control plane and not part of the engine: it keeps using the buffer
    pointer it peeked before popping, so the FIRST buffer is delivered twice;
    and it offsets the char* destination by a sample count, so a read spanning
    several buffers is garbage.  Hence: single-buffer reads, duplicate dropped."""
    R = ref()
    R.audio_init()
    out = []
    for k, b in enumerate(blocks):
        b = np.ascontiguousarray(b, dtype=np.int32).reshape(-1, 2)
        R.audio_fm_demodulator(_p(b), b.shape[0])
        buf = np.zeros(b.shape[0] // 4, dtype=np.float32)
        assert R.audio_get_audio_payload(_p(buf), buf.nbytes) == buf.nbytes
        if k == 1:                       # the replay of the first buffer
            assert np.array_equal(buf, out[0])
            assert R.audio_get_audio_payload(_p(buf), buf.nbytes) == buf.nbytes
        out.append(buf.copy())
    return out
