#!/usr/bin/env python3
"""Per-call latency of the drop-in entry points (host buffers in and out, PCIe
both ways, one synchronisation per call) on the GPU box -- the compatibility
path, never bench.py's value."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import rtlws                         # noqa: E402
from rtlws import synth              # noqa: E402


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e6 * (time.perf_counter() - t0) / n


iq = synth.tone_noise_iq(8, 1024, seed=1)
for N in (1024, 4096):
    s = rtlws.Spectrum(N)
    frame = synth.tone_noise_iq(1, N, seed=2)[0]
    ps = np.zeros(N)
    print("spectrum_add_cmplx_u8  N=%-5d %8.1f us/call" % (N, timeit(lambda: s.add_cmplx_u8(frame, ps), 2000)))
    s.free()
# the decimator's output fed to a spectrum (SURVEY.md §8a row a3 / configs[3]'s composition through the
# reference API: rf_decimator -> spectrum_add_cmplx_s32, src/spectrum.c:65-81) and the real-f32 entry
# point: one launch of the fused f64 kernel on the cmplx_s32 / f32 input kinds (round 4), or of the
# row-per-workgroup kernel with RTLWS_F64_FUSED=0
rng = np.random.default_rng(5)
for N in (1024, 2048, 4096):
    s = rtlws.Spectrum(N)
    s32 = rng.integers(-1024, 1024, size=(N, 2), dtype=np.int32)
    f32 = rng.standard_normal(N).astype(np.float32)
    ps = np.zeros(N)
    print("spectrum_add_cmplx_s32 N=%-5d %8.1f us/call" % (N, timeit(lambda: s.add_cmplx_s32(s32, ps), 2000)))
    print("spectrum_add_real_f32  N=%-5d %8.1f us/call" % (N, timeit(lambda: s.add_real_f32(f32, ps), 2000)))
    s.free()
blk = synth.uniform_iq(1, 204800, seed=3).reshape(-1, 2)       # 100 ms at 2.048 MS/s
print("cic_decimate R=10, 204800 samples   %8.1f us/call" % timeit(lambda: rtlws.cic_decimate(10, blk), 200))
blk8 = synth.uniform_iq(1, 153600, seed=3).reshape(-1, 2)
print("cic_decimate R=8, 153600 samples    %8.1f us/call" % timeit(lambda: rtlws.cic_decimate(8, blk8), 200))
x = np.random.default_rng(0).standard_normal(20480).astype(np.float32)
d = np.zeros(10, dtype=np.float32)
print("halfband_decimate 20480 -> 10240    %8.1f us/call" % timeit(lambda: rtlws.halfband_decimate(x, d), 500))
