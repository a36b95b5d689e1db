#!/usr/bin/env python3
"""Accuracy report (run on the GPU box): HIP engine vs f64 oracle, per-bin
relative error under several floors eps (|d| / max(|ref|, eps*max_bin))."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import rtlws                         # noqa: E402
from rtlws import synth              # noqa: E402
from oracle import pyoracle as po    # noqa: E402

eng = rtlws.Engine(0)
for N, K, nfr in ((1024, 1, 4096), (1024, 6, 4096 * 6 // 6), (2048, 1, 2048), (4096, 8, 2048)):
    for name, gen in (("tone+noise", synth.tone_noise_iq), ("pure tone", synth.pure_tone_iq),
                      ("uniform", synth.uniform_iq)):
        nframes = (nfr // K) * K
        iq = gen(nframes, N, seed=17)
        got = eng.spectra(iq, N, k_avg=K).astype(np.float64)
        ref = po.batch_spectra_u8(iq, N, K=K, nthreads=16)
        d = np.abs(got - ref)
        mx = ref.max(axis=1, keepdims=True)
        line = "N=%d K=%d %-10s" % (N, K, name)
        for eps in (1e-9, 1e-7, 1e-5):
            r = d / np.maximum(ref, eps * mx)
            line += " | eps=%g max %.2e p99.9 %.2e" % (eps, r.max(), np.quantile(r, 0.999))
        print(line, flush=True)
        if K == 1:
            # yardstick: scipy's single-precision pocketfft (complex64 in, complex64 arithmetic)
            # on the same input -- numpy's complex64 FFT computes in f64 and is no yardstick
            x = ((iq[..., 0].astype(np.float32) - 128) / 128 + 1j * (iq[..., 1].astype(np.float32) - 128) / 128).astype(np.complex64)
            import scipy.fft
            X = scipy.fft.fft(x, axis=-1)
            assert X.dtype == np.complex64
            P = (X.real.astype(np.float64) ** 2 + X.imag.astype(np.float64) ** 2)
            P = np.roll(P, N // 2, axis=-1)
            P[:, N // 2] = P[:, N // 2 - 1]
            dn = np.abs(P - ref)
            line = "      scipy f32 pocketfft  "
            for eps in (1e-9, 1e-7, 1e-5):
                r = dn / np.maximum(ref, eps * mx)
                line += " | eps=%g max %.2e p99.9 %.2e" % (eps, r.max(), np.quantile(r, 0.999))
            print(line, flush=True)
