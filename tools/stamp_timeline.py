#!/usr/bin/env python3
"""Timeline of one launch of the fused kernel (diagnostic).

Needs a library built with -DRTLWS_STAMP (make -C rtl-ws_amd variant NAME=stamp
EXTRA=-DRTLWS_STAMP), selected with RTLWS_HIP_LIB: every workgroup overwrites
the head of each row it produces with the wall-clock time (100 MHz) at which the
row was complete, and the head of its last row with {end, start, first row
complete, rows done, dispatch index}.  Prints when workgroups start, store
their first row and end, how many rows each took, and the chip's row rate over
time."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import rtlws      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 65536 * 1024 // N
dev = torch.device("cuda", 0)
eng = rtlws.Engine(0)
stream = rtlws.torch_stream_handle()
desc = rtlws.make_desc(N)
src = [torch.randint(0, 256, (frames, N, 2), dtype=torch.uint8, device=dev) for _ in range(3)]
dst = [torch.empty((frames, N), dtype=torch.float32, device=dev) for _ in range(3)]
for i in range(12):
    eng.spectra_batch(desc, src[i % 3].data_ptr(), frames, dst[i % 3].data_ptr(), stream=stream)
torch.cuda.synchronize()


def q(x):
    return "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f" % (
        x.min(), *np.percentile(x, [10, 50, 90, 99]), x.max())


def analyse(buf, label, t_ref=None):
    head = buf[:, :10].cpu().numpy().copy().view(np.uint64)          # 5 words per row
    last = head[:, 1] != 0                                             # a workgroup's last row
    wg = head[last]
    t0 = wg[:, 1].min() if t_ref is None else t_ref
    us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0
    end, start, first = us(wg[:, 0]), us(wg[:, 1]), us(wg[:, 2])
    rows, b = wg[:, 3].astype(int), wg[:, 4].astype(int)
    done = np.concatenate([us(head[~last, 0]), end])
    print("%s: %d workgroups  (microseconds)" % (label, len(wg)))
    print("  start       ", q(start))
    print("  first store ", q(first))
    print("  end         ", q(end))
    print("  rows per workgroup: min %d  p10 %d  p50 %d  p90 %d  max %d" % (
        rows.min(), *np.percentile(rows, [10, 50, 90]).astype(int), rows.max()))
    edges = np.arange(np.floor(done.min() / 4.0) * 4.0, done.max() + 4.0, 4.0)
    hist, _ = np.histogram(done, bins=edges)
    print("  rows completed per 4 us bin from %.0f us:" % edges[0], " ".join("%d" % h for h in hist))
    quarter = b * 4 // (b.max() + 1)
    print("  by dispatch quarter: end p50", " ".join("%.1f" % np.median(end[quarter == k]) for k in range(4)),
          " rows p50", " ".join("%d" % np.median(rows[quarter == k]) for k in range(4)))
    return wg[:, 1].min(), wg[:, 0].max()


# one launch on an idle chip
eng.spectra_batch(desc, src[0].data_ptr(), frames, dst[0].data_ptr(), stream=stream)
torch.cuda.synchronize()
analyse(dst[0], "isolated launch (after the first workgroup's start)")

# steady state: nine launches back to back, the last three survive in the three buffers
for i in range(9):
    eng.spectra_batch(desc, src[i % 3].data_ptr(), frames, dst[i % 3].data_ptr(), stream=stream)
torch.cuda.synchronize()
s6, e6 = analyse(dst[0], "back to back, launch 7 of 9 (after its first workgroup's start)")
s7, e7 = analyse(dst[1], "back to back, launch 8 of 9 (after launch 7's first start)", t_ref=s6)
s8, e8 = analyse(dst[2], "back to back, launch 9 of 9 (after launch 7's first start)", t_ref=s6)
print("launch period 7->8 %.2f us, 8->9 %.2f us;  last end -> next first start: %.2f, %.2f us" % (
    (float(s7) - float(s6)) / 100.0, (float(s8) - float(s7)) / 100.0,
    (float(s7) - float(e6)) / 100.0, (float(s8) - float(e7)) / 100.0))
