#!/usr/bin/env python3
"""Accuracy report (run on the GPU box): HIP engine vs f64 oracle, per-bin
relative error under several floors eps (|d| / max(|ref|, eps*max_bin))."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
        for eps in (1e-9, 1e-8, 1e-7, 1e-6):
            r = d / np.maximum(ref, eps * mx)
            line += " | eps=%g max %.2e p99.9 %.2e" % (eps, r.max(), np.quantile(r, 0.999))
        print(line, flush=True)
