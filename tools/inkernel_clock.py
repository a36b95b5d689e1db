#!/usr/bin/env python3
"""The shader clock a fused-kernel workload actually runs at, measured IN the kernel
(MI355X_MICROARCH.md, "DVFS give-back" (6)): every workgroup of the -DRTLWS_STAMP diagnostic build
stamps s_memtime (shader clocks) and s_memrealtime (100 MHz) at its start and end; after >= 2 s of
back-to-back launches on the bench's synthetic input the last launch's stamps give
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup; the median is printed.
rocm-smi's sclk (tools/power_ab.sh) can read up to ~10 % above this.

usage (GPU box):  make -C rtl-ws_amd variant NAME=stamp EXTRA=-DRTLWS_STAMP
                  RTLWS_HIP_LIB=rtl-ws_amd/lib/variants/stamp/librtlws_hip.so python tools/inkernel_clock.py [workload ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import bench      # noqa: E402
import rtlws      # noqa: E402

os.environ["RTLWS_V2"] = "0"           # the stamps live in spectrum_fused.hip
dev = torch.device("cuda", 0)
eng = rtlws.Engine(0)
side = torch.cuda.Stream(device=dev)
out = {}
for name in (sys.argv[1:] or ["batched_1024pt_64k_frames", "hann_4096pt_k8_db"]):
    n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
    assert output in ("power_sum", "mean_db") and name not in bench.F64_WORKLOADS, "f32 rows carry the stamps"
    desc = rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0)
    rows = frames // k_avg
    with torch.cuda.stream(side):
        ins = [bench.synth_iq_torch(torch, frames, n_fft * max(cic_r, 1), 1234 + 17 * s, dev) for s in range(4)]
        outs = [torch.empty((rows, n_fft), dtype=torch.float32, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    t0, i = time.time(), 0
    while time.time() - t0 < 2.5:                       # >= 2 s of back-to-back launches
        for _ in range(500):
            eng.spectra_batch(desc, ins[i % 4].data_ptr(), frames, outs[i % 4].data_ptr(), stream=side.cuda_stream)
            i += 1
        torch.cuda.synchronize()
    last = outs[(i - 1) % 4]
    head = last[:, :12].cpu().numpy().copy().view(np.uint64)          # 6 words per row
    wg = head[head[:, 1] != 0]                                         # a workgroup's last row
    real = (wg[:, 0] - wg[:, 1]).astype(np.float64)                    # 100 MHz ticks
    clk = wg[:, 5].astype(np.float64)
    ok = (real > 1000) & (clk > 0)
    mhz = 100.0 * clk[ok] / real[ok]
    out[name] = {"workgroups": int(ok.sum()), "launches": i, "sclk_mhz_median": float(np.median(mhz)),
                 "sclk_mhz_p10": float(np.percentile(mhz, 10)), "sclk_mhz_p90": float(np.percentile(mhz, 90)),
                 "workgroup_us_median": float(np.median(real[ok]) / 100.0)}
    print(name, json.dumps(out[name]), flush=True)
    del ins, outs
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "inkernel_clock.json"), "w"), indent=1, sort_keys=True)
