#!/usr/bin/env python3
"""Fused CIC + N-pt spectrum rates for several R (device-resident).

    python tools/cic_fused_rates.py [N] [R ...]

RTLWS_CIC_DIRECT=1 in the environment keeps R != 8 on the per-lane loads
(A/B against the LDS-staged input stage; run both in one gpurun call)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import rtlws      # noqa: E402

dev = torch.device("cuda", 0)
eng = rtlws.Engine(0)
L = rtlws.hip_lib()
stream = rtlws.torch_stream_handle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
RS = [int(a) for a in sys.argv[2:]] or [8, 10, 12, 5, 16]
print('N=%d  RTLWS_CIC_DIRECT=%s' % (N, os.environ.get('RTLWS_CIC_DIRECT', '0')))
for R in RS:
    nspec = 8192 * 8 * 2048 // (R * N)
    desc = rtlws.make_desc(N, cic_r=R)
    src = [torch.randint(0, 256, (nspec, N * R, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
    dst = [torch.empty((nspec, N), dtype=torch.float32, device=dev) for _ in range(3)]
    for i in range(5):
        eng.spectra_batch(desc, src[i % 3].data_ptr(), nspec, dst[i % 3].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = L.rtlws_event_create(), L.rtlws_event_create()
    steps = 60
    L.rtlws_event_record(e0, eng.h, stream)
    for i in range(steps):
        eng.spectra_batch(desc, src[i % 3].data_ptr(), nspec, dst[i % 3].data_ptr(), stream=stream)
    L.rtlws_event_record(e1, eng.h, stream)
    us = 1e3 * L.rtlws_event_elapsed_ms(e0, e1) / steps
    byts = nspec * (2 * N * R + 4 * N)
    print("R=%-3d spectra %6d  %8.1f us  %6.0f GB/s algorithmic  %.3e spectra/s" % (R, nspec, us, byts / us / 1e3, nspec / us * 1e6))
