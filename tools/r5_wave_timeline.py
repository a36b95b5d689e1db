#!/usr/bin/env python3
"""Per-wavefront timeline of one launch of spectrum_f64_1024x.hip (diagnostic).

Needs a library built with -DRTLWS_X_STAMP (make -C rtl-ws_amd variant NAME=x_stamp
EXTRA=-DRTLWS_X_STAMP) selected with RTLWS_HIP_LIB: lane 0 of every wavefront overwrites the
head of the last row it produced with {start, end (100 MHz), start, end (shader clocks),
HW_ID, XCC_ID, rows, workgroup}.  Prints, per SIMD, how long its co-resident wavefronts
lived -- do the two wavefronts of a SIMD share its vector pipe, or does the older one run
ahead and leave the younger one to finish alone?

usage (GPU box): RTLWS_HIP_LIB=.../x_stamp/librtlws_hip.so python3 tools/r5_wave_timeline.py [blocks_per_cu ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import rtlws      # noqa: E402

N, FRAMES = 1024, 65536
dev = torch.device("cuda", 0)
eng = rtlws.Engine(0)
stream = rtlws.torch_stream_handle()
desc = rtlws.make_desc(N, flags=rtlws.FLAG_ROWS_F32)
src = [torch.randint(0, 256, (FRAMES, N, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
dst = [torch.empty((FRAMES, N), dtype=torch.float32, device=dev) for _ in range(3)]


def q(x):
    return "min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f" % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())


def analyse(buf, label):
    head = buf[:, :16].cpu().numpy().copy().view(np.uint64)          # 8 words per row
    # a stamped row: word 6 (rows) in 1..65536 and word 7 (workgroup) < 65536 and word 0 < word 1
    ok = (head[:, 6] >= 1) & (head[:, 6] <= FRAMES) & (head[:, 7] < 65536) & (head[:, 0] < head[:, 1]) & (head[:, 4] < (1 << 32))
    w = head[ok]
    t0 = w[:, 0].min()
    start = (w[:, 0] - t0) / 100.0
    end = (w[:, 1] - t0) / 100.0
    clk = (w[:, 3] - w[:, 2]).astype(np.float64)
    ghz = clk / ((w[:, 1] - w[:, 0]).astype(np.float64) * 10.0)
    hw = w[:, 4].astype(np.int64)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = w[:, 5].astype(np.int64) & 15
    rows = w[:, 6].astype(np.int64)
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print("%s: %d wavefronts on %d SIMDs of %d CUs; launch %.1f us from first start to last end" % (
        label, len(w), len(np.unique(key)), len(np.unique(key >> 2)), end.max()))
    print("  start (us)       ", q(start))
    print("  end (us)         ", q(end))
    print("  lifetime (us)    ", q(end - start))
    print("  lifetime (kcycles)", q(clk / 1e3))
    print("  clock (GHz)      ", q(ghz))
    print("  rows per wavefront", q(rows.astype(np.float64)))
    print("  cycles per row    ", q(clk / rows))
    # per SIMD: sort its wavefronts by end time
    order = np.argsort(key, kind="stable")
    ks, idx = np.unique(key[order], return_index=True)
    groups = np.split(order, idx[1:])
    per = {}
    for g in groups:
        per.setdefault(len(g), []).append(g)
    for n in sorted(per):
        gs = per[n]
        e = np.array([np.sort(end[g]) for g in gs])
        s = np.array([np.sort(start[g]) for g in gs])
        life = np.array([np.sort(clk[g]) for g in gs]) / 1e3
        print("  SIMDs with %d wavefronts: %d" % (n, len(gs)))
        for k in range(n):
            print("    #%d to finish: end p50 %6.1f us (p10 %6.1f p90 %6.1f); lifetime p50 %6.1f kcycles; #%d to start: p50 %5.2f us" % (
                k + 1, np.median(e[:, k]), *np.percentile(e[:, k], [10, 90]), np.median(life[:, k]), k + 1, np.median(s[:, k])))
        # does the first-started wavefront finish first?
        first_started_first = np.mean([np.argmin(start[g]) == np.argmin(end[g]) for g in gs])
        print("    first to start is first to finish on %.0f %% of them" % (100 * first_started_first))
    return end.max()


# arguments: workgroups per CU of the one-wavefront form (1 .. 16), or w8 for the eight-wavefront workgroups
for arg in sys.argv[1:] or ["8"]:
    if arg.startswith("w"):
        eng.set_option("f64_x_waves", int(arg[1:]))
        per_cu = arg
    else:
        eng.set_option("f64_x_waves", 1)
        per_cu = int(arg)
        eng.set_option("f64_blocks_per_cu", per_cu)
    for i in range(600):      # settle the clock governor
        eng.spectra_batch_f64(desc, src[i % 3].data_ptr(), FRAMES, dst[i % 3].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    analyse(dst[(600 - 1) % 3], "f64_blocks_per_cu %s, steady state (launch 600 of 600)" % per_cu)
    eng.spectra_batch_f64(desc, src[0].data_ptr(), FRAMES, dst[0].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    analyse(dst[0], "f64_blocks_per_cu %s, isolated launch" % per_cu)
