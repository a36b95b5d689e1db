#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run from the repo root IN THE BUILD CONTAINER
(needs /root/reference for the reference-object-code vectors):

    python oracle/gen_golden.py

Two kinds of vector, told apart by the `source` field in each file:

  source = "reference_object_code"
      Outputs of the reference's own src/resample.c / src/rf_decimator.c,
      compiled unmodified into oracle/_ref/librtlws_ref.so (oracle/Makefile).
      These pin the oracle (and, through it, the HIP kernels) bit for bit.
        cic_ref.npz       cic_decimate R in {8,10,12}, incl. chained calls
                          (delay-line carry) and a hand-set delay state
        rfdec_ref.npz     rf_decimator re-blocking with odd chunk sizes
        halfband_ref.npz  halfband_decimate, two consecutive calls
        audio_ref.npz     audio_fm_demodulator (atan2_approx, difference, limiter,
                          two half-bands): four consecutive decimator blocks

  source = "oracle_f64"
      Outputs of our own f64 restatement (src/spectrum.c and src/cbb_main.c
      cannot be built here: FFTW3 is absent).  They are regression pins of
      the oracle, NOT reference-pinned vectors.
        spectrum_oracle.npz   1024-pt K=1 / K=6, 4096-pt K=8, 2048-pt on CIC
                              R=8 output, Hann 4096 (extension), edge cases
        payload_oracle.npz    dB payload bytes for gains {0,15,-25},
                              blocks {0,2,6}

  source = "fftw3_via_scipy_test_data"
      fftw_dct2_dst2.npz: outputs of the real FFTW3 library (REDFT10 / RODFT10 of
      x = 0..n-1, n in {2,3,4,8,12,15,16,17,32,...,1024}), extracted from the
      reference data SciPy ships for its own tests
      (scipy/fftpack/tests/fftw_double_ref.npz, produced by its gen_fftw_ref.py
      with FFTW).  FFTW3 is the library src/spectrum.c:21,42 calls and is absent
      from this image; these are the only FFTW-computed numbers in it.  They pin
      the oracle's DFT -- magnitude AND the forward sign convention, through the
      sine transform -- by the identities in tests/test_oracle_cpu.py.

Only data (inputs + expected outputs) is written; no reference source text.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))

from oracle import pyoracle as po  # noqa: E402
from rtlws import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_reference_vectors():
    if not po.ref_available():
        raise SystemExit("oracle/_ref/librtlws_ref.so missing: run `make -C oracle` "
                         "in the container that has /root/reference")
    rng = np.random.default_rng(20261003)

    # --- cic_decimate ---------------------------------------------------
    d = {"source": "reference_object_code"}
    for R in (8, 10, 12):
        src = rng.integers(0, 256, size=(R * 96, 2), dtype=np.uint8)
        # three chained calls of unequal length, state carried
        cuts = [0, R * 16, R * 56, R * 96]
        st = None
        outs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            rc, dst, st = po.ref_cic_decimate(R, src[a:b], state=st)
            assert rc == 0
            outs.append(dst)
        d[f"R{R}_src"] = src
        d[f"R{R}_cuts"] = np.array(cuts, dtype=np.int32)
        d[f"R{R}_dst"] = np.concatenate(outs)
        d[f"R{R}_state"] = st
    # hand-set delay line with comb_prev_in != integrator_prev_out
    src = rng.integers(0, 256, size=(8 * 8, 2), dtype=np.uint8)
    st0 = np.array([1000, -2000, 300, 40], dtype=np.int32)
    rc, dst, st = po.ref_cic_decimate(8, src, state=st0)
    d["odd_src"], d["odd_state0"], d["odd_dst"], d["odd_state1"] = src, st0, dst, st
    # int32 wrap of the running integrator
    st0 = np.array([2147483000, -2147483000, 2147483000, -2147483000], dtype=np.int32)
    src = np.full((8 * 64, 2), 255, dtype=np.uint8)
    src[:, 1] = 0
    rc, dst, st = po.ref_cic_decimate(8, src, state=st0)
    d["wrap_src"], d["wrap_state0"], d["wrap_dst"], d["wrap_state1"] = src, st0, dst, st
    # length mismatch -> -1, nothing written
    rc, _, _ = po.ref_cic_decimate(8, src[:63], dst_len=8)
    d["mismatch_rc"] = np.int32(rc)
    np.savez_compressed(os.path.join(OUT, "cic_ref.npz"), **d)

    # --- rf_decimator ---------------------------------------------------
    d = {"source": "reference_object_code"}
    fs, R = 48000.0, 8          # 100 ms block = 600 out / 4800 in: small fixture
    total = 4800 * 3 + 1234
    src = rng.integers(0, 256, size=(total, 2), dtype=np.uint8)
    cuts = [0, 1000, 1001, 5801, 9999, 10000, 14400, total]
    chunks = [src[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    rcs, blocks = po.ref_rf_decimate(fs, R, chunks)
    d["fs"], d["R"] = np.float64(fs), np.int32(R)
    d["src"], d["cuts"] = src, np.array(cuts, dtype=np.int32)
    d["rcs"] = np.array(rcs, dtype=np.int32)
    d["blocks"] = np.stack(blocks)
    np.savez_compressed(os.path.join(OUT, "rfdec_ref.npz"), **d)

    # --- halfband -------------------------------------------------------
    d = {"source": "reference_object_code"}
    x = rng.standard_normal(2 * 300).astype(np.float32)
    delay = np.zeros(10, dtype=np.float32)
    y1 = po.ref_halfband_decimate(x[:2 * 100], delay)
    delay_mid = delay.copy()
    y2 = po.ref_halfband_decimate(x[2 * 100:], delay)
    d["x"], d["y1"], d["y2"], d["delay_mid"], d["delay_end"] = x, y1, y2, delay_mid, delay.copy()
    np.savez_compressed(os.path.join(OUT, "halfband_ref.npz"), **d)


def gen_audio_vectors():
    """FM front end (src/audio_main.c + src/common_sp.h atan2_approx), reference
    object code.  The reference keeps state in function statics: this must be
    the only audio call of the process (it is)."""
    rng = np.random.default_rng(20261004)
    n = 2048 * 4
    t = np.arange(n) / 204800.0
    ph = 2 * np.pi * 40000 * np.cumsum(np.sin(2 * np.pi * 1000 * t)) / 204800.0
    iq = np.stack([800 * np.cos(ph), 800 * np.sin(ph)], axis=1) + rng.normal(0, 30, size=(n, 2))
    iq = np.round(iq).astype(np.int32)
    # every branch of atan2_approx: x == 0 (y >, ==, < 0), the axes, all quadrants, |z| on both sides of 1
    special = [(0, 7), (0, 0), (0, -3), (-5, 0), (5, 0), (3, 4), (-3, 4), (-3, -4), (3, -4),
               (4, 3), (-4, 3), (-4, -3), (4, -3), (1, 1), (-1, 1), (-1, -1), (1, -1)]
    for i, (re, im) in enumerate(special):
        iq[40 + i] = (re, im)
    blocks = [iq[i * 2048:(i + 1) * 2048] for i in range(4)]
    audio = po.ref_audio_chain(blocks)
    d = {"source": "reference_object_code", "iq": iq, "block_len": np.int32(2048),
         "audio": np.stack(audio)}
    np.savez_compressed(os.path.join(OUT, "audio_ref.npz"), **d)


def gen_oracle_vectors():
    d = {"source": "oracle_f64"}
    iq = synth.tone_noise_iq(6, 1024, seed=11)
    d["n1024_iq"] = iq
    d["n1024_k1"] = po.batch_spectra_u8(iq, 1024, K=1)
    d["n1024_k6"] = po.batch_spectra_u8(iq, 1024, K=6)
    iq = synth.tone_noise_iq(8, 4096, seed=12)
    d["n4096_iq"] = iq
    d["n4096_k8"] = po.batch_spectra_u8(iq, 4096, K=8)
    d["n4096_k8_hann"] = po.batch_spectra_u8(iq, 4096, K=8, window=synth.hann(4096))
    iq = synth.tone_noise_iq(1, 2048 * 8, seed=13)
    d["cic2048_iq"] = iq
    d["cic2048_k1"] = po.batch_spectra_cic_u8(iq, 2048, 8, K=1)
    # edge cases: all-128 (every bin 0), full-scale 0/255 square wave
    flat = np.full((1, 1024, 2), 128, dtype=np.uint8)
    d["flat_k1"] = po.batch_spectra_u8(flat, 1024)
    sq = np.zeros((1, 1024, 2), dtype=np.uint8)
    sq[0, ::2, 0] = 255
    sq[0, 1::2, 1] = 255
    d["square_iq"] = sq
    d["square_k1"] = po.batch_spectra_u8(sq, 1024)
    np.savez_compressed(os.path.join(OUT, "spectrum_oracle.npz"), **d)

    d = {"source": "oracle_f64"}
    sig = synth.tone_noise_iq(1, 131072, seed=21).reshape(-1, 2)
    d["iq_first6k"] = sig[: 6 * 1024 + 500]
    for blocks_len, tag in ((500, "b0"), (2 * 1024 + 952, "b2"), (6 * 1024 + 500, "b6")):
        ps, blocks = po.estimate_spectrum(sig[:blocks_len])
        d[f"{tag}_len"] = np.int32(blocks_len)
        d[f"{tag}_blocks"] = np.int32(blocks)
        for g in (0, 15, -25):
            d[f"{tag}_gain{g}"] = po.spectrum_payload(ps, blocks, g)
    np.savez_compressed(os.path.join(OUT, "payload_oracle.npz"), **d)


def gen_fftw_vectors():
    """FFTW-computed DCT-II / DST-II vectors out of SciPy's own test data (see the
    module docstring); skipped when this SciPy build does not ship them."""
    import scipy.fftpack
    src = os.path.join(os.path.dirname(scipy.fftpack.__file__), "tests", "fftw_double_ref.npz")
    if not os.path.exists(src):
        print("scipy fftw_double_ref.npz not found: keeping tests/golden/fftw_dct2_dst2.npz")
        return
    ref = np.load(src)
    d = {"source": "fftw3_via_scipy_test_data", "sizes": np.asarray(ref["sizes"])}
    for n in ref["sizes"]:
        d["dct2_%d" % n] = ref["dct_2_%d" % n]
        d["dst2_%d" % n] = ref["dst_2_%d" % n]
    np.savez_compressed(os.path.join(OUT, "fftw_dct2_dst2.npz"), **d)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_fftw_vectors()
    po.build()
    gen_reference_vectors()
    gen_audio_vectors()
    gen_oracle_vectors()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
