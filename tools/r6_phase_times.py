#!/usr/bin/env python3
"""Where a frame of spectrum_f64_4096y.hip spends its shader clocks (diagnostic; needs the -DRTLWS_Y_STAMP build:
make -C rtl-ws_amd yvariant NAME=y_stamp EXTRA=-DRTLWS_Y_STAMP, RTLWS_HIP_LIB=rtl-ws_amd/lib/variants/y_stamp/librtlws_hip.so).
Every wavefront sums, over its frames, the clocks between phase boundaries and leaves the sums in the head of an
output row; this prints the mean per frame and phase over all wavefronts, and the spread of the wavefronts' lifetimes.

usage (GPU box): python3 tools/r6_phase_times.py [workload]        (RTLWS_F64_BLOCKS_PER_CU=1: one workgroup per CU)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import rtlws            # noqa: E402
import bench            # noqa: E402

PHASES = ["read samples, next copy, convert, window, pass 1", "barrier 1", "exchange-1 writes + barrier 2", "exchange-1 reads + pass 2",
          "exchange 2 (writes, reads)", "pass 3 + |X|^2", "row epilogue (per row)", "wait for the samples' copy"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hann_4096pt_k8_db_f64c_f32o"
    n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
    prec = bench.precision_of(name)
    dev = torch.device("cuda", 0)
    eng = rtlws.Engine(0)
    stream = rtlws.torch_stream_handle()
    # R6_SETS=4: four rotating input sets (HBM-streamed, as bench.py); default 1 (the 128 MiB stay in the Infinity Cache)
    sets = int(os.environ.get("R6_SETS", "1"))
    srcs = [torch.randint(0, 256, (frames, n_fft, 2), dtype=torch.uint8, device=dev) for _ in range(sets)]
    src = srcs[0]
    desc = rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0, rtlws.FLAG_ROWS_F32 if prec == "f64c_f32o" else 0)
    odt = torch.float64 if prec == "f64" else torch.float32
    nrows = frames // k_avg
    dst = torch.zeros((nrows + 64, n_fft), dtype=odt, device=dev)        # the stamp records go behind the last row
    dst2 = torch.zeros_like(dst)
    for i in range(300):
        eng.spectra_batch_f64(desc, srcs[i % sets].data_ptr(), frames, dst.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        eng.spectra_batch_f64(desc, srcs[i % sets].data_ptr(), frames, dst.data_ptr(), stream=stream)
    e1.record()
    # two consecutive launches into two buffers: the gap between the first one's last wavefront and the second one's first
    eng.spectra_batch_f64(desc, srcs[1 % sets].data_ptr(), frames, dst.data_ptr(), stream=stream)
    eng.spectra_batch_f64(desc, srcs[2 % sets].data_ptr(), frames, dst2.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    print("HIP events: %.2f us per launch over 200 back-to-back launches, %d input set(s)" % (e0.elapsed_time(e1) * 1e3 / 200, sets))
    cus = eng.get_option("cu_count")
    per_cu = int(os.environ.get("RTLWS_F64_BLOCKS_PER_CU", "2") or 2)
    nwaves = cus * per_cu * 4
    raw = dst[nrows:].cpu().numpy().reshape(-1).view(np.uint64)[:12 * nwaves].reshape(nwaves, 12)
    ok = raw[:, 11] == 0x5354414d50          # records a later row store did not overwrite
    st = raw[ok].astype(np.float64)
    nfr = st[:, 8]
    print("%s, %d workgroup(s) per CU: %d of %d wavefront records intact, %.1f frames each" % (name, per_cu, st.shape[0], nwaves, nfr.mean()))
    tot = 0.0
    for i, ph in enumerate(PHASES):
        per = st[:, i] / nfr
        tot += per.mean()
        print("  %-40s %8.0f shader clocks per frame (min %6.0f, max %6.0f over the wavefronts)" % (ph, per.mean(), per.min(), per.max()))
    life = (st[:, 10] - st[:, 9]) / 100.0
    print("  %-40s %8.0f; wavefront lifetime %.1f us mean, %.1f min, %.1f max; first start to last end %.1f us" % (
        "sum", tot, life.mean(), life.min(), life.max(), (st[:, 10].max() - st[:, 9].min()) / 100.0))
    raw2 = dst2[nrows:].cpu().numpy().reshape(-1).view(np.uint64)[:12 * nwaves].reshape(nwaves, 12)
    st2 = raw2[raw2[:, 11] == 0x5354414d50].astype(np.float64)
    if st2.shape[0]:
        print("  next launch: its first wavefront's stamp comes %.1f us after this launch's last wavefront ended, %.1f us after its first started" % (
            (st2[:, 9].min() - st[:, 10].max()) / 100.0, (st2[:, 9].min() - st[:, 9].min()) / 100.0))
        starts = np.sort(st2[:, 9]) - st2[:, 9].min()
        print("  start stamps of the next launch's wavefronts, us after the first: median %.1f, 90 %% %.1f, last %.1f" % (
            np.median(starts) / 100.0, np.percentile(starts, 90) / 100.0, starts[-1] / 100.0))


main()
