#!/usr/bin/env python3
"""Energy per launch of ANY bench.py workload (diagnostic): the package energy accumulator of the device
(found by PCI bus id, rtl-ws_amd/rtlws/energy.py) and HIP-event time around a long run of back-to-back launches
over 4 rotating buffer sets, after 1 500 settle launches.

usage (GPU box): python3 tools/energy_per_launch.py <workload> [launches]   (kernel switches: RTLWS_* / RTLWS_HIP_LIB)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))
import torch      # noqa: E402
import rtlws      # noqa: E402
from rtlws import energy      # noqa: E402
import bench      # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else bench.HEADLINE
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
    prec = bench.precision_of(name)
    sets = 4
    dev = torch.device("cuda", 0)
    eng = rtlws.Engine(0)
    ec = energy.for_hip_device(rtlws, 0)
    assert ec is not None, "no energy counter for device 0 (rocm_smi / bus id)"
    stream = rtlws.torch_stream_handle()
    spf = n_fft * max(cic_r, 1)
    src = [torch.randint(0, 256, (frames, spf, 2), dtype=torch.uint8, device=dev) for _ in range(sets)]
    desc = rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0, rtlws.FLAG_ROWS_F32 if prec == "f64c_f32o" else 0)
    odt = torch.uint8 if output == "payload_u8" else torch.float64 if prec == "f64" else torch.float32
    # (+ 64 rows of slack: the -DRTLWS_Y_STAMP diagnostic builds leave their records behind the last row)
    dst = [torch.empty((frames // k_avg + 64, n_fft), dtype=odt, device=dev) for _ in range(sets)]
    fn = eng.spectra_batch if prec == "f32" else eng.spectra_batch_f64
    for i in range(1500):
        fn(desc, src[i % sets].data_ptr(), frames, dst[i % sets].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the shader clock: two stamp kernels in the stream around the launches (rtlws_clock_stamp; nothing resident beside
    # them).  R6_PROBE=1: the clock-probe wavefront of rounds 4-5 beside the launches instead -- it perturbs them, which
    # is what profiles/r06_clock_probe_perturbation.txt used it to show
    slots = 2048
    stamps = torch.zeros((2, slots, 4), dtype=torch.int64, device=dev)
    eng.clock_stamp(stamps[0].data_ptr(), slots, stream=stream)
    torch.cuda.synchronize()      # (before the probe exists: a device synchronise would wait for it to time out)
    probe = eng.clock_probe_start() if os.environ.get("R6_PROBE") else None
    j0, t0 = ec.joules(), time.time()
    eng.clock_stamp(stamps[0].data_ptr(), slots, stream=stream)
    e0.record()
    for i in range(launches):
        fn(desc, src[i % sets].data_ptr(), frames, dst[i % sets].data_ptr(), stream=stream)
    e1.record()
    eng.clock_stamp(stamps[1].data_ptr(), slots, stream=stream)
    if probe is not None:
        eng.clock_probe_signal_on_stream(probe, stream)
    torch.cuda.synchronize()
    j1, t1 = ec.joules(), time.time()
    sclk = eng.clock_probe_stop(probe)[0] if probe is not None else eng.clock_from_stamps(*stamps.cpu().numpy())[0]
    us = e0.elapsed_time(e1) * 1e3 / launches
    byt = bench.algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output, prec == "f64") * frames
    print("%-32s %-14s launches %d: %.2f us per launch (events) = %.4f of 8 TB/s, %.1f mJ per launch, %.0f W over %.2f s, "
          "sclk %.3f GHz -> %.1f k shader cycles per launch" % (
              name, os.environ.get("R5_LABEL", ""), launches, us, byt / (us * 1e-6) / 8e12,
              (j1 - j0) / launches * 1e3, (j1 - j0) / (t1 - t0), t1 - t0, sclk or 0.0, us * (sclk or 0.0)))


main()
