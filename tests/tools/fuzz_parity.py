#!/usr/bin/env python3
"""Randomised differential test (run on the GPU box): random descriptors and
batch sizes through rtlws_spectra_batch against the f64 oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtlws                         # noqa: E402
from rtlws import synth              # noqa: E402
from oracle import pyoracle as po    # noqa: E402
from helpers import rel_err          # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
eng = rtlws.Engine(0)
t0 = time.time()
n_cases = 0
worst = 0.0
while time.time() - t0 < budget_s:
    N = int(rng.choice([1024, 1024, 2048, 4096, 64, 100, 500, 768]))
    fused = N in (1024, 2048, 4096)
    K = int(rng.choice([1, 1, 2, 3, 6, 8, 17]))
    rows = int(rng.integers(1, 40 if fused else 4))
    window = str(rng.choice(["rect", "rect", "hann"]))
    cic_r = int(rng.choice([0, 0, 0, 2, 5, 8, 8, 12, 10, 7, 16, 24, 48])) if fused else int(rng.choice([0, 3]))
    out = str(rng.choice(["power_sum", "power_sum", "mean_db", "payload_u8"]))
    gain = int(rng.choice([0, 15, -25, 40]))
    gen = [synth.tone_noise_iq, synth.uniform_iq, synth.pure_tone_iq][int(rng.integers(0, 3))]
    w = None if window == "rect" else synth.hann(N)
    kind = str(rng.choice(["cu8", "cu8", "cu8", "cs32", "rf32"]))
    if kind != "cu8":                       # the s32 / real-f32 inputs of spectrum.h, row by row
        cic_r, rows = 0, min(rows, 6)
        if kind == "cs32":
            data = rng.integers(-4000, 4000, size=(rows * K, N, 2), dtype=np.int32)
        else:
            data = rng.standard_normal((rows * K, N)).astype(np.float32)
        got = eng.spectra(data, N, k_avg=K, input=kind, window=window, output=out, gain_db=gain)
        ref = np.zeros((rows, N))
        add = po.spectrum_add_cmplx_s32 if kind == "cs32" else po.spectrum_add_real_f32
        for r in range(rows):
            for k in range(K):
                assert add(N, data[r * K + k], ref[r], window=w) == 0
        gen = type("g", (), {"__name__": kind})
    else:
        iq = gen(rows * K, N * max(cic_r, 1), seed=int(rng.integers(0, 1 << 30)))
        got = eng.spectra(iq, N, k_avg=K, window=window, output=out, cic_r=cic_r, gain_db=gain)
    if kind != "cu8":
        pass
    elif cic_r > 1:
        ref = po.batch_spectra_cic_u8(iq, N, cic_r, K=K, window=w, nthreads=8)
    else:
        ref = po.batch_spectra_u8(iq, N, K=K, window=w, nthreads=8)
    tag = "N=%d K=%d rows=%d win=%s cic=%d out=%s gain=%d %s" % (N, K, rows, window, cic_r, out, gain, gen.__name__)
    if out == "power_sum":
        # bins further than this below the row maximum are judged in absolute terms
        # (f32 floor, DESIGN.md "Error budget": about 6e-8 of the peak AMPLITUDE, whatever K is)
        eps = 1e-5
        tol = 1e-4 if fused else 1e-3
        e = rel_err(got, ref, eps if fused else 1e-5).max()
        worst = max(worst, e if fused else 0.0)
        assert e <= tol, (tag, e)
    elif out == "mean_db":
        with np.errstate(divide="ignore"):
            want = 10 * np.log10(ref / K)
        mx = ref.max(axis=1, keepdims=True)
        ok = np.isfinite(want) & (ref > 1e-9 * mx)
        with np.errstate(divide="ignore", invalid="ignore"):
            tol_db = np.maximum(3e-4, 4.34 * 4e-7 * np.sqrt(mx / np.maximum(ref, 1e-300)))
        bad = ok & (np.abs(got - want) > (tol_db if fused else 10 * tol_db))
        assert not bad.any(), (tag, float(np.abs(got - want)[bad].max()))
    else:
        for r in range(rows):
            want = po.spectrum_payload(ref[r], K, gain)
            diff = got[r].astype(int) - want.astype(int)
            g = 10.0 ** (int(gain / 10))
            with np.errstate(divide="ignore", invalid="ignore"):
                d = 10 * np.log10(np.abs(g * ref[r] / K))
                # f32 error budget in dB for a bin of power p under a row maximum pmax:
                # 4.34 * (2 * 6e-8 * sqrt(pmax / p)), with a 3x margin; never below 2e-3
                tol_db = np.maximum(2e-3, 4.34 * 4e-7 * np.sqrt(ref[r].max() / np.maximum(ref[r], 1e-300)))
                near = np.abs(d - np.round(d)) < (tol_db if fused else 10 * tol_db)
            assert np.all((diff == 0) | (near & (np.abs(diff) == 1))), (tag, int(np.abs(diff).max()))
    n_cases += 1
print("fuzz seed %d: %d cases ok in %.1f s, worst fused power rel err %.2e" % (seed, n_cases, time.time() - t0, worst))
