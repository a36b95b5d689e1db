"""BASELINE.json configs[0]: boundary #2 (cbb_main.h) end to end over a
synthetic sensor -- a recorded-IQ replay behind rtl_sensor.h, the minimal
signal source, the GPU spectrum engine and the GPU dB/clamp kernel.
Reference behaviour being checked: src/cbb_main.c:40-70,106-135."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BUF_SAMPLES = 131072            # librtlsdr's default async buffer: 262144 bytes


def _run(tmp_path, built, iq, fs=2400000, speedup=1.0, seconds=2.0, gains=(0,)):
    rec = tmp_path / "iq.u8"
    iq.tofile(rec)
    os.environ["RTLWS_SYNTH_FILE"] = str(rec)
    os.environ["RTLWS_SYNTH_SPEEDUP"] = str(speedup)
    os.environ["RTLWS_SYNTH_BUFLEN"] = str(2 * BUF_SAMPLES)
    os.environ.pop("RTLWS_SYNTH_MAXBUFS", None)
    L = built.cbb_lib()
    L.cbb_init(192000)
    import ctypes as C
    synth_lib = C.CDLL(built.SYNTH_LIB, mode=C.RTLD_GLOBAL)
    synth_lib.rtl_set_sample_rate.argtypes = [C.c_void_p, C.c_uint32]
    synth_lib.rtl_set_sample_rate(L.cbb_get_rtl_dev(), fs)       # what main.c's "bw 2400" does
    updates, payloads = [], []
    t0 = time.time()
    try:
        while time.time() - t0 < seconds:
            if L.cbb_new_spectrum_available():
                updates.append(time.time() - t0)
                payloads.append([built.cbb_payload(g) for g in gains])
                assert L.cbb_new_spectrum_available() == 0        # flag cleared, src/cbb_main.c:132
            time.sleep(0.002)
        seen = L.rtlws_cbb_samples_seen()
    finally:
        L.cbb_close()
    return updates, payloads, seen


def test_live_path_payload_and_cadence(tmp_path, built, oracle):
    from rtlws import synth
    # one sensor buffer, replayed: every estimate sees the same first 6 frames
    iq = synth.tone_noise_iq(1, BUF_SAMPLES, seed=3).reshape(-1, 2)
    updates, payloads, seen = _run(tmp_path, built, iq, gains=(0, 15, -25))
    # 2.4 MS/s, 131072-sample buffers (54.6 ms), 250 ms gate -> every 5th buffer: ~3.7 Hz
    assert 3 <= len(updates) <= 9        # ~7 expected in 2 s; slack for a cold first launch
    if len(updates) >= 3:
        gaps = np.diff(updates)
        assert 0.2 < np.median(gaps) < 0.4
    assert seen >= 10 * BUF_SAMPLES
    ps, blocks = oracle.estimate_spectrum(iq[: 6 * 1024 + 100])
    assert blocks == 6
    for got in payloads:
        for g, gain in zip(got, (0, 15, -25)):
            want = oracle.spectrum_payload(ps, 6, gain)
            assert g.size == 1024
            assert np.array_equal(g, want)          # f64 sums + f64 dB: identical bytes


def test_short_buffers_two_blocks_and_none(tmp_path, built, oracle):
    from rtlws import synth
    # 3000-sample buffers: blocks = 2 (src/cbb_main.c:44,49)
    iq = synth.tone_noise_iq(1, 3000, seed=4).reshape(-1, 2)
    rec = tmp_path / "iq.u8"
    iq.tofile(rec)
    os.environ["RTLWS_SYNTH_FILE"] = str(rec)
    os.environ["RTLWS_SYNTH_SPEEDUP"] = "0.02"      # 3000 samples per 62 ms
    os.environ["RTLWS_SYNTH_BUFLEN"] = str(2 * 3000)
    L = built.cbb_lib()
    L.cbb_init(192000)
    try:
        t0 = time.time()
        while not L.cbb_new_spectrum_available() and time.time() - t0 < 3:
            time.sleep(0.002)
        assert L.cbb_new_spectrum_available() == 1
        got = built.cbb_payload(0)
        ps, blocks = oracle.estimate_spectrum(iq)
        assert blocks == 2 and got.size == 1024
        assert np.array_equal(got, oracle.spectrum_payload(ps, 2, 0))
        # buf_len is honoured (the reference ignores it)
        assert built.cbb_payload(0, buf_len=100).size == 100
    finally:
        L.cbb_close()
    # buffers shorter than one frame: flag set, payload empty (count 0)
    os.environ["RTLWS_SYNTH_BUFLEN"] = str(2 * 500)
    os.environ["RTLWS_SYNTH_SPEEDUP"] = "0.005"
    L.cbb_init(192000)
    try:
        t0 = time.time()
        while not L.cbb_new_spectrum_available() and time.time() - t0 < 3:
            time.sleep(0.002)
        assert L.cbb_new_spectrum_available() == 1
        assert built.cbb_payload(0).size == 0
    finally:
        L.cbb_close()


def test_decimator_is_fed_by_cbb(tmp_path, built):
    """cbb_rf_decimator() is the decimator the audio chain hooks into
    (src/main.c:205); R = fs / 192000 with integer division (src/cbb_main.c:80)."""
    import ctypes as C
    from rtlws import synth
    iq = synth.uniform_iq(1, BUF_SAMPLES, seed=9).reshape(-1, 2)
    rec = tmp_path / "iq.u8"
    iq.tofile(rec)
    os.environ["RTLWS_SYNTH_FILE"] = str(rec)
    os.environ["RTLWS_SYNTH_SPEEDUP"] = "4"
    os.environ["RTLWS_SYNTH_BUFLEN"] = str(2 * BUF_SAMPLES)
    L = built.cbb_lib()
    L.cbb_init(192000)
    blocks = []

    @C.CFUNCTYPE(None, C.c_void_p, C.c_int)
    def cb(ptr, n):
        blocks.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 2)).copy())

    try:
        built.amd_lib().rf_decimator_add_callback(L.cbb_rf_decimator(), C.cast(cb, C.c_void_p))
        time.sleep(0.6)
    finally:
        L.cbb_close()
    assert len(blocks) >= 2
    # synthetic sensor default fs = 2.048 MS/s -> R = 10, 100 ms blocks of 20480 outputs
    assert all(b.shape == (20480, 2) for b in blocks)
    stream = np.concatenate([iq] * 8)[: 204800]
    want = (stream.astype(np.int32) - 128).reshape(-1, 10, 2).sum(axis=1)
    assert np.array_equal(blocks[0], want)


def test_reference_cbb_main_object_code_over_gpu_engine(tmp_path, built, oracle):
    """Boundary #1 as the reference itself uses it: the reference's unmodified
    src/cbb_main.c + src/signal_source.c (object code built by oracle/Makefile into
    oracle/_ref/rtlws_ref_cbb_on_gpu) call spectrum_alloc / spectrum_add_cmplx_u8 /
    rf_decimator_* of librtlws_amd.so through their own headers.  The payload's
    dB arithmetic is the reference's f64 code; only the power sums are ours."""
    import subprocess
    from rtlws import synth
    exe = os.path.join(os.path.dirname(built.ROOT), "oracle", "_ref", "rtlws_ref_cbb_on_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/rtlws_ref_cbb_on_gpu not built (needs /root/reference at build time)")
    iq = synth.tone_noise_iq(1, BUF_SAMPLES, seed=8).reshape(-1, 2)
    rec = tmp_path / "iq.u8"
    iq.tofile(rec)
    env = dict(os.environ, RTLWS_SYNTH_FILE=str(rec), RTLWS_SYNTH_SPEEDUP="1.0",
               RTLWS_SYNTH_BUFLEN=str(2 * BUF_SAMPLES))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
    ps, blocks = oracle.estimate_spectrum(iq[: 6 * 1024 + 10])
    lines = [l for l in out.stdout.splitlines() if l.startswith("gain ")]
    assert len(lines) == 3
    for line, gain in zip(lines, (0, 15, -25)):
        parts = line.split()
        assert int(parts[1]) == gain and int(parts[3]) == 1024
        got = np.frombuffer(bytes.fromhex(parts[4]), dtype=np.uint8)
        want = oracle.spectrum_payload(ps, 6, gain)
        assert np.array_equal(got, want)            # reference dB code over our f64 sums


def test_all_frames_mode(tmp_path, built, oracle):
    """RTLWS_CBB_ALL_FRAMES=1: all 128 frames of a sensor buffer in
    one launch, payload format unchanged."""
    from rtlws import synth
    iq = synth.tone_noise_iq(1, BUF_SAMPLES, seed=6).reshape(-1, 2)
    os.environ["RTLWS_CBB_ALL_FRAMES"] = "1"
    try:
        updates, payloads, _ = _run(tmp_path, built, iq, seconds=1.0, gains=(0,))
    finally:
        os.environ.pop("RTLWS_CBB_ALL_FRAMES", None)
    assert len(updates) >= 2
    ref = oracle.batch_spectra_u8(iq, 1024, K=128, nthreads=8)[0]
    want = oracle.spectrum_payload(ref, 128, 0)
    for (got,) in payloads:
        assert got.size == 1024 and np.array_equal(got, want)


def test_welch_mode_every_buffer_of_the_interval(tmp_path, built, oracle):
    """RTLWS_CBB_ALL_FRAMES=2: every sensor buffer is transformed (all 128 frames of it)
    and the 250 ms gate only publishes the running average.  The replayed recording is one
    buffer, so an interval of m buffers is the same 128 frames m times over, and the
    published row must be what the reference's sequential loop leaves after 128*m frames
    -- slot N/2 included, whose weights depend on a frame's position in the WHOLE sequence
    (rtlws_welch_accumulate_f64 / rtlws_welch_finish_f64)."""
    from rtlws import synth
    iq = synth.tone_noise_iq(1, BUF_SAMPLES, seed=12).reshape(-1, 2)
    rec = tmp_path / "iq.u8"
    iq.tofile(rec)
    os.environ["RTLWS_SYNTH_FILE"] = str(rec)
    os.environ["RTLWS_SYNTH_SPEEDUP"] = "1.0"
    os.environ["RTLWS_SYNTH_BUFLEN"] = str(2 * BUF_SAMPLES)
    os.environ.pop("RTLWS_SYNTH_MAXBUFS", None)
    os.environ["RTLWS_CBB_ALL_FRAMES"] = "2"
    L = built.cbb_lib()
    L.rtlws_cbb_published_frames.restype = __import__("ctypes").c_int
    got = []
    try:
        L.cbb_init(192000)
        import ctypes as C
        synth_lib = C.CDLL(built.SYNTH_LIB, mode=C.RTLD_GLOBAL)
        synth_lib.rtl_set_sample_rate.argtypes = [C.c_void_p, C.c_uint32]
        synth_lib.rtl_set_sample_rate(L.cbb_get_rtl_dev(), 2400000)
        t0 = time.time()
        while time.time() - t0 < 1.6:
            if L.cbb_new_spectrum_available():
                frames = L.rtlws_cbb_published_frames()
                got.append((frames, built.cbb_payload(0), built.cbb_payload(15)))
            time.sleep(0.002)
    finally:
        L.cbb_close()
        os.environ.pop("RTLWS_CBB_ALL_FRAMES", None)
    assert len(got) >= 3
    # 2.4 MS/s, 54.6 ms buffers, 250 ms gate: 4-5 buffers = 512-640 frames per interval
    # (the first interval may be shorter or longer: cold start)
    assert all(f % 128 == 0 and f >= 128 for f, _, _ in got)
    assert any(f >= 512 for f, _, _ in got)
    cache = {}
    for frames, p0, p15 in got:
        if frames not in cache:
            m = frames // 128
            cache[frames] = oracle.batch_spectra_u8(np.tile(iq, (m, 1)), 1024, K=frames, nthreads=8)[0]
        ref = cache[frames]
        assert np.array_equal(p0, oracle.spectrum_payload(ref, frames, 0))
        assert np.array_equal(p15, oracle.spectrum_payload(ref, frames, 15))
