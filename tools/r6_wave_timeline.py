#!/usr/bin/env python3
"""Per-wavefront timeline of one launch of a spectrum_f64_fused.hip workload (diagnostic; VERDICT r5 item 3: where the
power cap does NOT bind -- cic8_2048pt_f64 -- does the younger wavefront of a SIMD finish alone for a large share of
the launch, as it did in spectrum_f64_1024x before round 5?).

Needs a library built with -DRTLWS_F_STAMP (make -C rtl-ws_amd fvariant NAME=f_stamp EXTRA=-DRTLWS_F_STAMP) selected
with RTLWS_HIP_LIB: lane 0 of every wavefront leaves {start, end (100 MHz), start, end (shader clocks), HW_ID, XCC_ID,
rows, id} behind the last output row (this tool allocates the room).

usage (GPU box): RTLWS_HIP_LIB=.../f_stamp/librtlws_hip.so python3 tools/r6_wave_timeline.py [workload ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import rtlws      # noqa: E402
import bench      # noqa: E402

dev = torch.device("cuda", 0)
eng = rtlws.Engine(0)
stream = rtlws.torch_stream_handle()
MAGIC = 0x5354414d5


def q(x):
    return "min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f" % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())


def run(name):
    n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
    prec = bench.precision_of(name)
    assert prec != "f32", "the stamp hook lives in spectrum_f64_fused.hip"
    spf = n_fft * max(cic_r, 1)
    rows = frames // k_avg
    src = [torch.randint(0, 256, (frames, spf, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
    desc = rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0, rtlws.FLAG_ROWS_F32 if prec == "f64c_f32o" else 0)
    odt = torch.float64 if prec == "f64" else torch.float32
    elt = 8 if prec == "f64" else 4
    slack_rows = (1 << 20) // (n_fft * elt) + 2              # 1 MiB for the records
    dst = [torch.zeros((rows + slack_rows, n_fft), dtype=odt, device=dev) for _ in range(3)]
    for i in range(600):      # settle the clock governor
        eng.spectra_batch_f64(desc, src[i % 3].data_ptr(), frames, dst[i % 3].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(300):
        eng.spectra_batch_f64(desc, src[i % 3].data_ptr(), frames, dst[i % 3].data_ptr(), stream=stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 300
    rec = dst[(300 - 1) % 3][rows:].cpu().numpy().reshape(-1).view(np.uint64)
    rec = rec[:(rec.size // 8) * 8].reshape(-1, 8)
    w = rec[(rec[:, 7] >> 28) == MAGIC]
    t0 = w[:, 0].min()
    start = (w[:, 0] - t0) / 100.0
    end = (w[:, 1] - t0) / 100.0
    clk = (w[:, 3] - w[:, 2]).astype(np.float64)
    hw = w[:, 4].astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    xcc = w[:, 5].astype(np.int64) & 15
    nrows = w[:, 6].astype(np.float64)
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    launch = end.max()
    print("%s: %.2f us per launch by HIP events (stamp build); %d wavefronts on %d SIMDs of %d CUs; first start to last end %.1f us" % (
        name, us, len(w), len(np.unique(key)), len(np.unique(key >> 2)), launch))
    print("  start (us)        ", q(start))
    print("  end (us)          ", q(end))
    print("  lifetime (kcycles)", q(clk / 1e3))
    print("  clock (GHz)       ", q(clk / ((w[:, 1] - w[:, 0]).astype(np.float64) * 10.0)))
    print("  rows per wavefront", q(nrows))
    order = np.argsort(key, kind="stable")
    ks, idx = np.unique(key[order], return_index=True)
    alone = []
    per = {}
    for g in np.split(order, idx[1:]):
        e = np.sort(end[g])
        per.setdefault(len(g), []).append(e)
        if len(g) >= 2:
            alone.append((e[-1] - e[-2]) / launch)
    for n in sorted(per):
        e = np.array(per[n])
        print("  SIMDs with %d wavefronts: %d; their k-th wavefront to finish ends at (p50 us): %s" % (
            n, len(e), "  ".join("%.1f" % np.median(e[:, k]) for k in range(n))))
    alone = np.array(alone)
    print("  share of the launch a SIMD's LAST wavefront runs with no other of this launch beside it: mean %.1f %%, p50 %.1f %%, p90 %.1f %%" % (
        100 * alone.mean(), 100 * np.median(alone), 100 * np.percentile(alone, 90)))


for name in sys.argv[1:] or ["cic8_2048pt_f64"]:
    run(name)
