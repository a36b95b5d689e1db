"""FM audio front end on the GPU (include/audio_main.h, rtlws_fm_demod) against
the reference-object-code golden and the oracle: bit-exact f32."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _drain(L, nfloats):
    buf = np.zeros(nfloats, dtype=np.float32)
    n = L.audio_get_audio_payload(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    return buf[: n // 4]


def test_audio_chain_golden(built):
    g = golden("audio_ref.npz")
    n = int(g["block_len"])
    L = built.amd_lib()
    L.audio_get_audio_payload.argtypes = [C.c_void_p, C.c_int]
    L.audio_fm_demodulator.argtypes = [C.c_void_p, C.c_int]
    L.audio_init()
    try:
        assert L.audio_new_audio_available() == 0
        for k in range(g["audio"].shape[0]):
            blk = np.ascontiguousarray(g["iq"][k * n:(k + 1) * n])
            L.audio_fm_demodulator(blk.ctypes.data_as(C.c_void_p), n)
            assert L.audio_new_audio_available() == 1
            got = _drain(L, n // 4)
            assert np.array_equal(got, g["audio"][k])          # bit-exact, delay lines carried
            assert L.audio_new_audio_available() == 0
        # FIFO across buffers: two blocks queued, read in three uneven pieces
        for k in range(2):
            blk = np.ascontiguousarray(g["iq"][k * n:(k + 1) * n])
            L.audio_fm_demodulator(blk.ctypes.data_as(C.c_void_p), n)
        a = np.concatenate([_drain(L, 300), _drain(L, 500), _drain(L, 10000)])
        assert a.size == 2 * (n // 4) and L.audio_new_audio_available() == 0
    finally:
        L.audio_close()


def test_fm_demod_kernel_vs_oracle(engine, oracle, built):
    rng = np.random.default_rng(12)
    n = 1 << 18
    iq = rng.integers(-3000, 3000, size=(n, 2), dtype=np.int32)
    iq[rng.integers(0, n, 500), 0] = 0                       # x == 0 branches
    iq[rng.integers(0, n, 500), 1] = 0
    d_iq = engine.upload(iq)
    d_prev = engine.upload(np.array([0.25, 0.0], dtype=np.float32))
    d_out = engine.alloc(n * 4)
    rc = built.hip_lib().rtlws_fm_demod(engine.h, d_iq.ptr, n, d_prev.ptr, d_prev.ptr + 4, d_out.ptr, None)
    assert rc == 0
    got = engine.download(d_out, np.float32, (n,))
    carry = engine.download(d_prev, np.float32, (2,))
    want, prev = oracle.fm_demod(iq, prev_phase=0.25)
    assert np.array_equal(got, want)
    assert carry[1] == np.float32(prev)
    # in == out pointer is refused
    assert built.hip_lib().rtlws_fm_demod(engine.h, d_iq.ptr, n, d_prev.ptr, d_prev.ptr, d_out.ptr, None) == -1


def test_decimator_to_audio_pipeline(built, oracle):
    """The product wiring of src/main.c:205: rf_decimator -> audio_fm_demodulator,
    both on the GPU, against oracle CIC + oracle audio chain."""
    from rtlws import synth
    L = built.amd_lib()
    L.audio_get_audio_payload.argtypes = [C.c_void_p, C.c_int]
    fs, R = 81920.0, 8                     # 100 ms blocks: 8192 in -> 1024 out -> 256 audio
    iq = synth.tone_noise_iq(1, 8192 * 3, seed=77).reshape(-1, 2)
    L.audio_init()
    d = built.RfDecimator()
    try:
        fn = C.cast(L.audio_fm_demodulator, C.c_void_p)
        built.amd_lib().rf_decimator_add_callback(d.h, fn)
        assert d.set_parameters(fs, R) == 0
        assert d.decimate(iq) == 0
        got = _drain(L, 3 * 256)
    finally:
        d.free()
        L.audio_close()
    st = np.zeros(21, dtype=np.float32)
    want = []
    for b in range(3):
        rc, dec, _ = oracle.cic_decimate(R, iq[b * 8192:(b + 1) * 8192])
        want.append(oracle.audio_block(dec, st))
    assert np.array_equal(got, np.concatenate(want))


def test_pool_exhaustion_leaves_delay_line_2_untouched(built, oracle):
    """reference src/audio_main.c:137-142: when none of the 50 pool buffers is free the
    block is dropped BEFORE the second half-band, so delay line 2 keeps the state of the
    last queued block while the phase carry and delay line 1 (first half-band, :133) keep
    advancing.  50 blocks fill the pool, 3 more are dropped, the pool is drained, and the
    54th block must come out as the oracle computes it from exactly that mixed state."""
    rng = np.random.default_rng(5)
    n, pool = 512, 50
    blocks = [rng.integers(-2000, 2000, size=(n, 2), dtype=np.int32) for _ in range(pool + 4)]
    L = built.amd_lib()
    L.audio_get_audio_payload.argtypes = [C.c_void_p, C.c_int]
    L.audio_fm_demodulator.argtypes = [C.c_void_p, C.c_int]
    st = np.zeros(21, dtype=np.float32)
    want = []
    for k, b in enumerate(blocks):
        keep = st[11:21].copy()
        out = oracle.audio_block(b, st)
        if pool <= k < pool + 3:
            st[11:21] = keep                  # dropped: second half-band never ran
        else:
            want.append(out)
    L.audio_init()
    try:
        for b in blocks[:pool + 3]:
            L.audio_fm_demodulator(np.ascontiguousarray(b).ctypes.data_as(C.c_void_p), n)
        got = _drain(L, (pool + 3) * (n // 4))
        assert got.size == pool * (n // 4)                       # three blocks were dropped
        assert np.array_equal(got, np.concatenate(want[:pool]))
        L.audio_fm_demodulator(np.ascontiguousarray(blocks[pool + 3]).ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(_drain(L, n // 4), want[pool])
        # a second audio_init without audio_close restarts the queue only: the carried
        # state (function statics in the reference, :76-79) survives
        L.audio_fm_demodulator(np.ascontiguousarray(blocks[0]).ctypes.data_as(C.c_void_p), n)
        L.audio_init()
        assert L.audio_new_audio_available() == 0
        nxt = oracle.audio_block(blocks[0], st)                  # the queued-then-discarded block
        nxt = oracle.audio_block(blocks[1], st)
        L.audio_fm_demodulator(np.ascontiguousarray(blocks[1]).ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(_drain(L, n // 4), nxt)
    finally:
        L.audio_close()
