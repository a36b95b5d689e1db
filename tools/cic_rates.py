#!/usr/bin/env python3
"""Stand-alone CIC kernel rates for several R (device-resident, algorithmic bytes 2R+8 per output)."""
import os
import sys
import time

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
RS = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5, 8, 10, 12, 16, 25, 64, 127, 128]
for R in RS:
    n_out = (1 << 27) // max(R, 4)
    n_out -= n_out % 64
    src = [torch.randint(0, 256, (n_out * R, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
    dst = [torch.empty((n_out, 2), dtype=torch.int32, device=dev) for _ in range(3)]
    for i in range(5):
        eng.cic_block_sums(R, src[i % 3].data_ptr(), n_out, dst[i % 3].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = L.rtlws_event_create(), L.rtlws_event_create()
    steps = 50
    L.rtlws_event_record(e0, eng.h, stream)
    for i in range(steps):
        eng.cic_block_sums(R, src[i % 3].data_ptr(), n_out, dst[i % 3].data_ptr(), stream=stream)
    L.rtlws_event_record(e1, eng.h, stream)
    ms = L.rtlws_event_elapsed_ms(e0, e1)
    us = 1e3 * ms / steps
    want = (src[(steps - 1) % 3][: 1000 * R].cpu().numpy().astype(np.int32) - 128).reshape(-1, R, 2).sum(axis=1)
    ok = np.array_equal(dst[(steps - 1) % 3][:1000].cpu().numpy(), want)
    print("R=%-3d outputs %9d  %8.1f us  %6.0f GB/s algorithmic  %.3e in-samples/s  exact=%s"
          % (R, n_out, us, n_out * (2 * R + 8) / us / 1e3, n_out * R / us * 1e6, ok))
