#!/usr/bin/env python3
"""Energy per launch of the configs[1] kernels (diagnostic): package energy accumulator
(rocm_smi rsmi_dev_energy_count_get) and HIP-event time around a long run of back-to-back launches.

usage (GPU box): python3 tools/r5_energy.py [f64c_f32o|f64|f32] [launches]   (kernel switches: RTLWS_* / RTLWS_HIP_LIB)
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

# the accumulator of HIP device 0, found by its PCI bus id (rocm_smi ignores HIP_VISIBLE_DEVICES: ADVICE r5)
_EC = energy.for_hip_device(rtlws, 0)
assert _EC is not None, "no energy counter for device 0"


def energy_j():
    return _EC.joules()


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f64c_f32o"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
    N, FRAMES, SETS = 1024, 65536, 4
    dev = torch.device("cuda", 0)
    eng = rtlws.Engine(0)
    stream = rtlws.torch_stream_handle()
    src = [torch.randint(0, 256, (FRAMES, N, 2), dtype=torch.uint8, device=dev) for _ in range(SETS)]
    if mode == "f64":
        desc, odt, fn = rtlws.make_desc(N), torch.float64, eng.spectra_batch_f64
    elif mode == "f64c_f32o":
        desc, odt, fn = rtlws.make_desc(N, flags=rtlws.FLAG_ROWS_F32), torch.float32, eng.spectra_batch_f64
    else:
        desc, odt, fn = rtlws.make_desc(N), torch.float32, eng.spectra_batch
    dst = [torch.empty((FRAMES, N), dtype=odt, device=dev) for _ in range(SETS)]
    for i in range(1500):
        fn(desc, src[i % SETS].data_ptr(), FRAMES, dst[i % SETS].data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    j0, t0 = energy_j(), time.time()
    e0.record()
    for i in range(launches):
        fn(desc, src[i % SETS].data_ptr(), FRAMES, dst[i % SETS].data_ptr(), stream=stream)
    e1.record()
    torch.cuda.synchronize()
    j1, t1 = energy_j(), time.time()
    us = e0.elapsed_time(e1) * 1e3 / launches
    print("%-10s %s launches %d: %.2f us per launch (events), %.1f mJ per launch, %.0f W over %.2f s wall" % (
        mode, os.environ.get("R5_LABEL", ""), launches, us, (j1 - j0) / launches * 1e3, (j1 - j0) / (t1 - t0), t1 - t0))


main()
