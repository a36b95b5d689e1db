#!/usr/bin/env python3
"""Randomised differential test of the f64 kernel (run on the GPU box): random
descriptors through rtlws_spectra_batch_f64 -- the kernel behind spectrum_add_*
and cbb_main.h -- against the f64 oracle, strict metric (floor 1e-9) at 1e-10,
payload bytes identical."""
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
from helpers import rel_err, EPS_STRICT, TOL_F64          # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
eng = rtlws.Engine(0)
t0 = time.time()
n_cases, worst, bytes_checked = 0, 0.0, 0
while time.time() - t0 < budget_s:
    u = rng.random()
    fused_size = u < 0.35
    if fused_size:
        N = int(rng.choice([1024, 2048, 4096]))               # spectrum_f64_fused.hip (cmplx_u8, no CIC)
    elif u < 0.65:
        N = int(2 ** rng.integers(1, 14))                     # 2 .. 8192
    else:
        N = int(rng.integers(2, 1500))                        # direct-sum sizes, primes included
    K = int(rng.choice([1, 1, 2, 3, 6, 8]))
    rows = int(rng.integers(1, 40)) if fused_size else int(rng.integers(1, 4))
    window = str(rng.choice(["rect", "rect", "hann"]))
    out = str(rng.choice(["power_sum", "power_sum", "mean_db", "payload_u8"]))
    gain = int(rng.choice([0, 15, -25, 40, -9, 9]))
    kind = str(rng.choice(["cu8", "cu8", "cu8", "cs32", "rf32"]))
    cic_r = int(rng.choice([0, 0, 3, 8, 10, 12])) if (kind == "cu8" and N <= (4096 if fused_size else 2048)) else 0
    rows_f32 = bool(out != "payload_u8" and rng.random() < 0.3)      # RTLWS_FLAG_ROWS_F32: one f32 rounding
    w = None if window == "rect" else synth.hann(N)
    if kind == "cu8":
        gen = [synth.tone_noise_iq, synth.uniform_iq, synth.pure_tone_iq][int(rng.integers(0, 3))]
        data = gen(rows * K, N * max(cic_r, 1), seed=int(rng.integers(0, 1 << 30)))
        if cic_r > 1:
            ref = po.batch_spectra_cic_u8(data, N, cic_r, K=K, window=w)
        else:
            ref = po.batch_spectra_u8(data, N, K=K, window=w)
    else:
        if kind == "cs32":
            data = rng.integers(-100000, 100000, size=(rows * K, N, 2), dtype=np.int32)
        else:
            data = rng.standard_normal((rows * K, N)).astype(np.float32)
        ref = np.zeros((rows, N))
        add = po.spectrum_add_cmplx_s32 if kind == "cs32" else po.spectrum_add_real_f32
        for r in range(rows):
            for k in range(K):
                assert add(N, data[r * K + k], ref[r], window=w) == 0
    got = eng.spectra(data, N, k_avg=K, input=kind, window=window, output=out, cic_r=cic_r,
                      gain_db=gain, f64=True, rows_f32=rows_f32)
    tag = "N=%d K=%d rows=%d win=%s cic=%d out=%s gain=%d %s%s" % (N, K, rows, window, cic_r, out, gain, kind,
                                                                  " f32rows" if rows_f32 else "")
    if rows_f32:
        assert got.dtype == np.float32, tag
        if out == "power_sum":
            big = ref.max() < 3e38                                # (int32-extreme frames overflow an f32 row)
            e = rel_err(got, ref, EPS_STRICT).max() if big else 0.0
            assert e <= 6.0e-8, (tag, e)
        else:
            ok = ref > 1e-9 * ref.max(axis=1, keepdims=True)
            with np.errstate(divide="ignore"):
                want = 10 * np.log10(ref / K)
            assert np.abs(got - want)[ok].max() <= 2e-5, (tag, float(np.abs(got - want)[ok].max()))
    elif out == "power_sum":
        e = rel_err(got, ref, EPS_STRICT).max()
        worst = max(worst, e)
        assert e <= TOL_F64, (tag, e)
    elif out == "mean_db":
        mx = ref.max(axis=1, keepdims=True)
        ok = ref > 1e-9 * mx                                  # bins above the strict floor
        with np.errstate(divide="ignore"):
            want = 10 * np.log10(ref / K)
        assert np.abs(got - want)[ok].max() <= 1e-8, (tag, float(np.abs(got - want)[ok].max()))
    else:
        for r in range(rows):
            want = po.spectrum_payload(ref[r], K, gain)
            assert np.array_equal(got[r], want), (tag, int(np.abs(got[r].astype(int) - want.astype(int)).max()))
            bytes_checked += want.size
    n_cases += 1
print("f64 fuzz seed %d: %d cases ok in %.1f s, worst strict rel err %.2e, %d payload bytes identical"
      % (seed, n_cases, time.time() - t0, worst, bytes_checked))
