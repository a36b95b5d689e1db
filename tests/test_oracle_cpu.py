"""CPU suite: the oracle against mathematics, numpy and the golden fixtures.

Pinning status (DESIGN.md "Oracle"): CIC / half-band / re-blocker goldens are
outputs of the reference's own object code (source = reference_object_code);
spectrum / payload goldens are regression pins of the f64 oracle itself
(source = oracle_f64) -- src/spectrum.c cannot be built here (FFTW3 absent)."""
import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("N", [2, 4, 8, 16, 64, 256, 1024, 4096])
def test_dft_pow2_vs_direct_and_numpy(oracle, N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    a = oracle.dft(x)
    b = np.fft.fft(x)
    scale = np.abs(b).max()
    assert np.abs(a - b).max() / scale < 2e-15 * np.log2(N) + 1e-15
    if N <= 1024:
        d = oracle.dft_direct(x)
        assert np.abs(a - d).max() / scale < 2e-15 * np.log2(N) + 1e-15


@pytest.mark.parametrize("N", [3, 12, 100, 1000])
def test_dft_any_n(oracle, N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    assert np.abs(oracle.dft(x) - np.fft.fft(x)).max() / np.abs(x).sum() < 1e-14


def test_dft_is_forward_unnormalised(oracle):
    """exp(-2 pi i n k / N), no 1/N: a tone at +k0 lands in bin k0 with height N."""
    N, k0 = 64, 5
    x = np.exp(2j * np.pi * k0 * np.arange(N) / N)
    X = oracle.dft(x)
    assert abs(X[k0] - N) < 1e-10 and np.abs(np.delete(X, k0)).max() < 1e-10


def _numpy_restatement(iq, K, window=None):
    """src/spectrum.c:15-35,47-63 in numpy, sequential DC-slot rule included."""
    x = (iq[..., 0].astype(np.float64) - 128) / 128 + 1j * (iq[..., 1].astype(np.float64) - 128) / 128
    if window is not None:
        x = x * window
    P = np.abs(np.fft.fft(x, axis=-1)) ** 2
    N = x.shape[-1]
    out = np.zeros((iq.shape[0] // K, N))
    for g in range(out.shape[0]):
        for k in range(K):
            p = np.roll(P[g * K + k], N // 2)          # slot i shows bin (i + N/2) % N
            out[g, :N // 2] += p[:N // 2]
            out[g, N // 2] += out[g, N // 2 - 1]         # running neighbour, already updated
            out[g, N // 2 + 1:] += p[N // 2 + 1:]
    return out


@pytest.mark.parametrize("N,K", [(1024, 1), (1024, 6), (2048, 1), (4096, 8)])
def test_spectrum_vs_numpy_restatement(oracle, N, K):
    from rtlws import synth
    iq = synth.tone_noise_iq(2 * K, N, seed=N + K)
    got = oracle.batch_spectra_u8(iq, N, K=K)
    ref = _numpy_restatement(iq, K)
    assert (np.abs(got - ref) / np.abs(ref)).max() < 1e-11
    # tone lands where the fft-shift says: slot (round(f*N) + N/2) % N
    w = synth.hann(N)
    gotw = oracle.batch_spectra_u8(iq, N, K=K, window=w)
    refw = _numpy_restatement(iq, K, window=w)
    assert (np.abs(gotw - refw) / np.abs(refw).max()).max() < 1e-12


def test_dc_slot_closed_form(oracle):
    """K frames into a zeroed row: slot N/2 = sum_k (K-k) P_k[N-1]."""
    from rtlws import synth
    iq = synth.uniform_iq(6, 1024, seed=1)
    per = oracle.batch_spectra_u8(iq, 1024, K=1)
    six = oracle.batch_spectra_u8(iq, 1024, K=6)[0]
    want = sum((6 - k) * per[k][511] for k in range(6))
    assert abs(six[512] - want) / want < 1e-13
    # non-zero starting buffer adds ps0[N/2] + K*ps0[N/2-1]
    ps = np.full(1024, 2.0)
    for k in range(6):
        assert oracle.spectrum_add_cmplx_u8(1024, iq[k], ps) == 0
    assert abs(ps[512] - (want + 2.0 + 6 * 2.0)) / want < 1e-13


def test_spectrum_len_mismatch_and_inputs(oracle):
    ps = np.zeros(1024)
    iq = np.zeros((1000, 2), dtype=np.uint8)
    assert oracle.spectrum_add_cmplx_u8(1024, iq, ps, length=1000) == -1
    assert not ps.any()
    # s32: value/128 with no offset; f32: (x, 0)
    rng = np.random.default_rng(2)
    s = rng.integers(-500, 500, size=(64, 2), dtype=np.int32)
    ps = np.zeros(64)
    assert oracle.spectrum_add_cmplx_s32(64, s, ps) == 0
    X = np.fft.fft((s[:, 0] + 1j * s[:, 1]) / 128.0)
    want = np.roll(np.abs(X) ** 2, 32)
    want[32] = want[31]
    assert np.allclose(ps, want, rtol=1e-12)
    f = rng.standard_normal(64).astype(np.float32)
    ps = np.zeros(64)
    assert oracle.spectrum_add_real_f32(64, f, ps) == 0
    want = np.roll(np.abs(np.fft.fft(f.astype(np.float64))) ** 2, 32)
    want[32] = want[31]
    assert np.allclose(ps, want, rtol=1e-12)


def test_all_128_gives_all_zero(oracle):
    flat = np.full((1, 1024, 2), 128, dtype=np.uint8)
    assert not oracle.batch_spectra_u8(flat, 1024).any()


def test_spectrum_goldens(oracle):
    g = golden("spectrum_oracle.npz")
    assert str(g["source"]) == "oracle_f64"
    assert np.allclose(oracle.batch_spectra_u8(g["n1024_iq"], 1024), g["n1024_k1"], rtol=1e-12)
    assert np.allclose(oracle.batch_spectra_u8(g["n1024_iq"], 1024, K=6), g["n1024_k6"], rtol=1e-12)
    assert np.allclose(oracle.batch_spectra_u8(g["n4096_iq"], 4096, K=8), g["n4096_k8"], rtol=1e-12)
    assert np.allclose(oracle.batch_spectra_cic_u8(g["cic2048_iq"], 2048, 8), g["cic2048_k1"], rtol=1e-12)
    assert not g["flat_k1"].any()


# ---- decimators: pinned to the reference's own object code --------------------

@pytest.mark.parametrize("R", [8, 10, 12])
def test_cic_golden_reference_object_code(oracle, R):
    g = golden("cic_ref.npz")
    assert str(g["source"]) == "reference_object_code"
    src, cuts = g[f"R{R}_src"], g[f"R{R}_cuts"]
    st, outs = None, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, dst, st = oracle.cic_decimate(R, src[a:b], state=st)
        assert rc == 0
        outs.append(dst)
    assert np.array_equal(np.concatenate(outs), g[f"R{R}_dst"])
    assert np.array_equal(st, g[f"R{R}_state"])
    # and the closed form: int32 block sums of (x - 128)
    want = (src.astype(np.int32) - 128).reshape(-1, R, 2).sum(axis=1)
    assert np.array_equal(g[f"R{R}_dst"], want)


def test_cic_golden_state_and_errors(oracle):
    g = golden("cic_ref.npz")
    rc, dst, st = oracle.cic_decimate(8, g["odd_src"], state=g["odd_state0"])
    assert rc == 0 and np.array_equal(dst, g["odd_dst"]) and np.array_equal(st, g["odd_state1"])
    rc, dst, st = oracle.cic_decimate(8, g["wrap_src"], state=g["wrap_state0"])
    assert rc == 0 and np.array_equal(dst, g["wrap_dst"]) and np.array_equal(st, g["wrap_state1"])
    rc, _, _ = oracle.cic_decimate(8, g["wrap_src"][:63], dst_len=8)
    assert rc == int(g["mismatch_rc"]) == -1


def test_halfband_golden_reference_object_code(oracle):
    g = golden("halfband_ref.npz")
    assert str(g["source"]) == "reference_object_code"
    delay = np.zeros(10, dtype=np.float32)
    y1 = oracle.halfband_decimate(g["x"][:200], delay)
    assert np.array_equal(delay, g["delay_mid"])
    y2 = oracle.halfband_decimate(g["x"][200:], delay)
    assert np.array_equal(y1, g["y1"]) and np.array_equal(y2, g["y2"])
    assert np.array_equal(delay, g["delay_end"])


def test_rf_decimator_golden_reference_object_code(oracle):
    g = golden("rfdec_ref.npz")
    assert str(g["source"]) == "reference_object_code"
    d = oracle.RfDecimator()
    assert d.decimate(np.zeros((4, 2), dtype=np.uint8)) == -1          # unconfigured
    rcs = [d.set_parameters(float(g["fs"]), int(g["R"]))]
    assert (d.resampled_len, d.input_len) == (600, 4800)
    src, cuts = g["src"], g["cuts"]
    for a, b in zip(cuts[:-1], cuts[1:]):
        rcs.append(d.decimate(src[a:b]))
    assert rcs == list(g["rcs"])
    assert np.array_equal(np.stack(d.blocks), g["blocks"])
    d.close()


def test_rf_decimator_block_lengths(oracle):
    d = oracle.RfDecimator()
    for fs, R, want in ((1.536e6, 8, (19200, 153600)), (2.4e6, 12, (20000, 240000)),
                        (2.048e6, 10, (20480, 204800))):
        assert d.set_parameters(fs, R) == 0
        assert (d.resampled_len, d.input_len) == want
    assert d.set_parameters(0, 8) == -1 and d.set_parameters(1e6, 0) == -1
    d.close()


@pytest.mark.skipif("not __import__('oracle.pyoracle').pyoracle.ref_available()")
def test_oracle_vs_reference_object_code_live(oracle):
    """When oracle/_ref exists (it travels to the GPU box prebuilt), compare on
    fresh random inputs too, not only on the committed fixtures."""
    rng = np.random.default_rng(99)
    for R in (2, 5, 8, 16):
        src = rng.integers(0, 256, size=(R * 500, 2), dtype=np.uint8)
        st0 = rng.integers(-1000, 1000, size=4).astype(np.int32)
        a = oracle.cic_decimate(R, src, state=st0)
        b = oracle.ref_cic_decimate(R, src, state=st0)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    x = rng.standard_normal(2 * 777).astype(np.float32)
    d1 = rng.standard_normal(10).astype(np.float32)
    d2 = d1.copy()
    assert np.array_equal(oracle.halfband_decimate(x, d1), oracle.ref_halfband_decimate(x, d2))
    assert np.array_equal(d1, d2)


# ---- payload (src/cbb_main.c:106-135) -----------------------------------------

def test_payload_semantics(oracle):
    ps = np.array([0.0, 1.0, 10.0, 12.3, 1e30, 5e-7] + [100.0] * 1018)
    out = oracle.spectrum_payload(ps, 1, 0)
    assert list(out[:6]) == [0, 0, 10, 10, 255, 0]        # -inf->0, trunc, clamp, negative->0
    assert out[6] == 20
    # gain moves in 10 dB steps: 15 -> 10^1, -25 -> 10^-2 (C integer division)
    assert oracle.spectrum_payload(ps, 1, 15)[6] == 30
    assert oracle.spectrum_payload(ps, 1, -25)[6] == 0 and oracle.spectrum_payload(ps * 1e4, 1, -25)[6] == 40
    # count divides; count == 0 -> nothing written
    assert oracle.spectrum_payload(ps, 10, 0)[6] == 10
    assert oracle.spectrum_payload(ps, 0, 0).size == 0


def test_payload_goldens_and_estimate_blocks(oracle):
    g = golden("payload_oracle.npz")
    sig = g["iq_first6k"]
    for tag, blocks in (("b0", 0), ("b2", 2), ("b6", 6)):
        n = int(g[f"{tag}_len"])
        ps, b = oracle.estimate_spectrum(sig[:n])
        assert b == blocks == int(g[f"{tag}_blocks"])
        for gain in (0, 15, -25):
            want = g[f"{tag}_gain{gain}"]
            assert np.array_equal(oracle.spectrum_payload(ps, b, gain), want)
            assert want.size == (1024 if blocks else 0)


# ---- FM audio front end: pinned to the reference's object code --

def test_audio_golden_reference_object_code(oracle):
    g = golden("audio_ref.npz")
    assert str(g["source"]) == "reference_object_code"
    n = int(g["block_len"])
    st = np.zeros(21, dtype=np.float32)
    for k in range(g["audio"].shape[0]):
        out = oracle.audio_block(g["iq"][k * n:(k + 1) * n], st)
        assert np.array_equal(out, g["audio"][k])              # bit-exact f32, state carried


def test_atan2_approx_branches(oracle):
    import math
    # x == 0 (src/common_sp.h:47-56)
    assert oracle.atan2_approx(3, 0) == np.float32(math.pi / 2)
    assert oracle.atan2_approx(-3, 0) == -np.float32(math.pi / 2)
    assert oracle.atan2_approx(0, 0) == 0.0
    # within ~0.005 rad of atan2 everywhere else (it is an approximation)
    rng = np.random.default_rng(1)
    for y, x in rng.integers(-2000, 2000, size=(500, 2)):
        if x == 0:
            continue
        a, b = oracle.atan2_approx(y, x), math.atan2(y, x)
        err = abs(a - b)
        assert min(err, abs(err - 2 * math.pi)) < 6e-3


def test_fm_demod_difference_and_limit(oracle):
    iq = np.array([[100, 0], [0, 100], [-100, 0], [100, 1], [100, 2]], dtype=np.int32)
    out, prev = oracle.fm_demod(iq, prev_phase=0.0)
    assert out[0] == 0.0
    assert out[1] == 1.0 and out[2] == 1.0        # +pi/2 steps are limited to +1
    assert out[3] == -1.0                         # the wrap from +pi to ~0 is limited to -1
    assert abs(out[4] - 0.01) < 1e-3 and abs(prev - 0.02) < 1e-3


def test_header_atan2_approx_matches_the_pinned_restatement(tmp_path, oracle):
    """include/common_sp.h carries atan2_approx for reference units that stay in the build
    (src/audio_main.c); it must agree bit for bit with the oracle's, which is pinned to the
    reference's object code through tests/golden/audio_ref.npz."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "a.c"
    src.write_text('#include "common_sp.h"\nfloat hdr_atan2(float y, float x) { return atan2_approx(y, x); }\n'
                   'int hdr_macros(void) { cmplx_u8 a; cmplx_s32 s, t, r; set_cmplx_u8(a, 200, 3);\n'
                   '  set_cmplx_s32_cmplx_u8(s, a, -128); set_cmplx_s32(t, s); add_cmplx_s32(s, t, r);\n'
                   '  sub_cmplx_s32(r, t, r); return real_cmplx_s32(r) * 1000 + imag_cmplx_s32(r)\n'
                   '  + real_cmplx_u8(a) * 1000000 + imag_cmplx_u8(a) * 100000000; }\n')
    so = tmp_path / "a.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(root, "include"),
                    "-o", str(so), str(src), "-lm"], check=True)
    L = ctypes.CDLL(str(so))
    L.hdr_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
    L.hdr_atan2.restype = ctypes.c_float
    rng = np.random.default_rng(3)
    pts = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (5, 5), (-5, 5), (5, -5), (-5, -5), (1, -3), (-1, -3)]
    pts += [tuple(v) for v in rng.integers(-3000, 3000, size=(4000, 2))]
    for y, x in pts:
        a, b = L.hdr_atan2(float(y), float(x)), oracle.atan2_approx(float(y), float(x))
        assert np.float32(a).tobytes() == np.float32(b).tobytes(), (y, x, a, b)
    # (200-128) = 72, (3-128) = -125: s = (72,-125); r = s + s - s
    assert L.hdr_macros() == 72 * 1000 - 125 + 200 * 1000000 + 3 * 100000000


def test_oracle_dft_reproduces_fftw_computed_vectors(oracle):
    """The only FFTW3-computed numbers in this image: DCT-II / DST-II (REDFT10 / RODFT10)
    of x = 0..n-1 from SciPy's own test data (tests/golden/fftw_dct2_dst2.npz,
    oracle/gen_golden.py).  With z[2j+1] = x_j, z[4n-2j-1] = +-x_j (length 4n) the forward
    DFT gives  Re Z_k = 2 sum x_j cos(pi (j+1/2) k / n)      = REDFT10[k]      (even z)
               -Im Z_k = 2 sum x_j sin(pi (j+1/2) k / n)     = RODFT10[k-1]    (odd z)
    -- the second fixes the SIGN of the exponent (forward = exp(-2 pi i nk/N), what
    fftw_plan_dft_1d(..., FFTW_FORWARD, ...) of src/spectrum.c:42 computes): a +i DFT
    would return the sine transform negated.  4n runs to 4096: the oracle's radix-2 path
    and (n = 3, 12, 15, 17) its direct path."""
    g = golden("fftw_dct2_dst2.npz")
    assert str(g["source"]) == "fftw3_via_scipy_test_data"
    for n in g["sizes"]:
        n = int(n)
        x = np.linspace(0, n - 1, n)
        for name, sgn in (("dct2", 1.0), ("dst2", -1.0)):
            z = np.zeros(4 * n, dtype=np.complex128)
            z[1:2 * n:2] = x
            z[4 * n - 1:2 * n:-2] = sgn * x
            Z = oracle.dft(z)
            got = Z.real[:n] if name == "dct2" else -Z.imag[1:n + 1]
            want = g["%s_%d" % (name, n)]
            assert np.abs(got - want).max() <= 2e-15 * np.abs(want).max(), (name, n)
            # and the part that must vanish does (even z -> real spectrum, odd z -> imaginary)
            other = Z.imag if name == "dct2" else Z.real
            assert np.abs(other).max() <= 2e-13 * max(np.abs(want).max(), 1.0)
