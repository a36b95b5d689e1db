#!/usr/bin/env python3
"""bench.py -- spectra/s of the fused IQ -> power-spectrum hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: ONE launch of the fused
kernel over 65 536 frames of 1024 cmplx_u8 (BASELINE.json configs[1]: 128 MiB
of device-resident IQ in, 256 MiB of f32 power spectra out, K=1, rectangular
window = reference behaviour).  Inputs are resident in HBM before the timed
region; steps rotate over 4 buffer sets (1.5 GiB) so nothing is served from
the 256 MiB Infinity Cache.  With N > 1 every rank runs the same batch on its
own GPU (independent frames, no collective on the data path): weak scaling.
The kernel runs at the package power cap, whose clock governor needs ~25 ms to
settle: at least 500 untimed launches precede the timed region (W of them are
the warm-up steps; the rest are reported as `settle_launches`).

One JSON line on stdout (rank 0).  `roofline` prices the kernel against HBM
using the ALGORITHMIC bytes (2*N in + 4*N/K out per frame);
`cpu_baseline` times the f64 oracle (oracle/, kind "port") on this host's
cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rtl-ws_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
SETTLE_LAUNCHES = 500          # untimed launches before the timed region, at least

WORKLOADS = {
    # name: (n_fft, k_avg, window, output, cic_r, frames per step)
    "batched_1024pt_64k_frames": (1024, 1, "rect", "power_sum", 0, 65536),
    "hann_4096pt_k8_db": (4096, 8, "hann", "mean_db", 0, 16384),
    "cic8_2048pt": (2048, 1, "rect", "power_sum", 8, 8192),
    # the reference's own factor at BASELINE's 2.4 MS/s (src/main.c:23,154: 2.4e6 / 192e3 = 12)
    "cic12_2048pt": (2048, 1, "rect", "power_sum", 12, 5456),
    # stand-alone CIC (reference src/resample.c:6-45): "frame" = 2048 decimated outputs,
    # 2*8*2048 bytes in, 8*2048 bytes out; unit reported: decimated samples/s
    "cic8_block_sums": (2048, 1, "rect", "cs32", 8, 8192),
}


def algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output="power_sum"):
    if output == "cs32":                       # stand-alone CIC: cmplx_s32 out
        return 2 * n_fft * cic_r + 8 * n_fft
    return 2 * n_fft * max(cic_r, 1) + 4 * n_fft // k_avg


def synth_iq_torch(torch, nframes, samples_per_frame, seed, device):
    """Tone (amp 0.6, random frequency) + Gaussian noise (sigma 0.05), quantised
    to offset-binary u8 -- SURVEY.md §8d's input, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((nframes, samples_per_frame, 2), dtype=torch.uint8, device=device)
    chunk = 4096
    n = torch.arange(samples_per_frame, device=device, dtype=torch.float32)[None, :]
    for a in range(0, nframes, chunk):
        b = min(a + chunk, nframes)
        f = torch.rand((b - a, 1), generator=g, device=device) - 0.5
        ph = torch.rand((b - a, 1), generator=g, device=device) * 6.283185307179586
        arg = 6.283185307179586 * torch.remainder(f * n, 1.0) + ph
        re = 0.6 * torch.cos(arg) + 0.05 * torch.randn((b - a, samples_per_frame), generator=g, device=device)
        im = 0.6 * torch.sin(arg) + 0.05 * torch.randn((b - a, samples_per_frame), generator=g, device=device)
        out[a:b, :, 0] = torch.clamp(torch.round(re * 128 + 128), 0, 255).to(torch.uint8)
        out[a:b, :, 1] = torch.clamp(torch.round(im * 128 + 128), 0, 255).to(torch.uint8)
    return out


def init_distributed(torch, backend="nccl", device=None):
    """One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE/
    MASTER_*).  The data path never uses the group: it only carries the timing
    barrier and the MAX-reduce of the elapsed time.  Returns (dist or None, world, rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # under torch.distributed.run a 1-rank job still forms its group, so N = 1 and
    # N > 1 go through the same code
    if world <= 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        return None, 1, 0
    import torch.distributed as dist
    # the contract is ONE line on stdout: the box exports NCCL_DEBUG=VERSION, which makes
    # RCCL print a version banner there; drop exactly that setting (anything else stays)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=device)    # nccl == RCCL on ROCm
    else:
        dist.init_process_group(backend=backend)
    return dist, world, rank


def max_over_ranks(torch, dist, values, device=None):
    """Element-wise MAX of a list of floats over all ranks (identity when dist is None)."""
    if dist is None:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def whole_job_rate(world, steps, frames_per_step, elapsed_s):
    """Every rank processes its own `frames_per_step` per step (weak scaling)."""
    return world * steps * frames_per_step / elapsed_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500,
                    help="untimed launches first; the clock governor needs ~300 (25 ms) to settle at the power cap")
    ap.add_argument("--workload", default="batched_1024pt_64k_frames", choices=sorted(WORKLOADS))
    ap.add_argument("--sets", type=int, default=4, help="rotating buffer sets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=0, help="override frames per step (experiments)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rtlws

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist, world, rank = init_distributed(torch, "nccl", device)

    n_fft, k_avg, window, output, cic_r, frames = WORKLOADS[args.workload]
    if args.frames > 0:
        frames = args.frames - args.frames % k_avg
    spf = n_fft * max(cic_r, 1)
    eng = rtlws.Engine(local_rank)
    cic_only = (output == "cs32")
    desc = None if cic_only else rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0)
    rows = frames // k_avg
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run); "
              "reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)

    # device-resident inputs / outputs, allocated by torch (plumbing only)
    ins = [synth_iq_torch(torch, frames, spf, 1234 + 17 * s + 1000 * rank, device) for s in range(args.sets)]
    out_dtype = torch.int32 if cic_only else torch.float32
    out_cols = 2 * n_fft if cic_only else n_fft
    outs = [torch.empty((rows, out_cols), dtype=out_dtype, device=device) for _ in range(args.sets)]
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        s = i % args.sets
        if cic_only:
            eng.cic_block_sums(cic_r, ins[s].data_ptr(), frames * n_fft, outs[s].data_ptr(), stream=stream)
        else:
            eng.spectra_batch(desc, ins[s].data_ptr(), frames, outs[s].data_ptr(), stream=stream)

    # First collective = RCCL's lazy communicator set-up (~16 ms): do it here, not
    # between the warm-up launches and the timed region, where that much idle
    # time would put the timed launches back into the governor's transient.
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()

    # The kernels run at the package power cap and the clock governor needs
    # ~300 launches (25 ms) to settle: whatever W is, at least
    # SETTLE_LAUNCHES untimed launches precede the timed region.
    settle = max(0, SETTLE_LAUNCHES - args.warmup)
    for i in range(settle + args.warmup):
        step(i)
    torch.cuda.synchronize()

    ev0, ev1 = rtlws.hip_lib().rtlws_event_create(), rtlws.hip_lib().rtlws_event_create()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rtlws.hip_lib().rtlws_event_record(ev0, eng.h, stream)
    for i in range(args.steps):
        step(i)
    rtlws.hip_lib().rtlws_event_record(ev1, eng.h, stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ev_ms = rtlws.hip_lib().rtlws_event_elapsed_ms(ev0, ev1)

    elapsed, ev_ms = max_over_ranks(torch, dist, [elapsed, ev_ms], device)

    result = None
    if rank == 0:
        value = whole_job_rate(world, args.steps, frames, elapsed)
        bytes_per_launch = algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output) * frames
        avg_launch_s = (ev_ms / 1e3) / args.steps
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        if cic_only:
            value *= n_fft                      # decimated samples per second
        result = {
            "metric": ("decimated samples/s (CIC R=%d)" % cic_r) if cic_only else
                      ("spectra/s (1024-pt IQ frames)" if n_fft == 1024 else "spectra/s (%d-pt IQ frames)" % n_fft),
            "value": value,
            "unit": "samples/s" if cic_only else "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_launches": settle,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32" if cic_only else "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "n_fft": n_fft, "frames_per_step": frames,
                       "k_avg": k_avg, "window": window, "output": output, "cic_r": cic_r,
                       "input": "cmplx_u8 tone(0.6)+noise(0.05), device-resident, %d rotating sets" % args.sets,
                       "sharding": "independent frames per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": 1e6 * avg_launch_s},
        }

        # parity spot check of what was just timed (first 256 frames of set 0)
        from oracle import pyoracle as po
        nchk = 256 * k_avg
        host_in = ins[0][:nchk].cpu().numpy()
        got = outs[0][:256].cpu().numpy().astype(np.float64)
        if cic_only:
            want = (host_in.astype(np.int32) - 128).reshape(-1, cic_r, 2).sum(axis=1).reshape(nchk, -1)
            result["parity"] = {"frames": nchk, "bit_exact": bool(np.array_equal(outs[0][:nchk].cpu().numpy(), want))}
            ref = None
        elif cic_r > 1:
            ref = po.batch_spectra_cic_u8(host_in, n_fft, cic_r, K=k_avg, nthreads=8)
        else:
            ref = po.batch_spectra_u8(host_in, n_fft, K=k_avg, nthreads=8,
                                      window=None if window == "rect" else
                                      (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)))
        if ref is None:
            pass
        elif output == "mean_db":
            ref = 10 * np.log10(ref / k_avg)
            result["parity"] = {"max_abs_db_err": float(np.abs(got - ref).max()), "frames": nchk}
        else:
            mx = ref.max(axis=1, keepdims=True)
            result["parity"] = {
                "frames": nchk,
                "max_rel_err_floor1e-5": float((np.abs(got - ref) / np.maximum(ref, 1e-5 * mx)).max()),
                "max_rel_err_floor1e-9": float((np.abs(got - ref) / np.maximum(ref, 1e-9 * mx)).max()),
            }

        if world == 1 and not args.no_cpu_baseline:
            # the GPU box gives one GPU's job a 16-CPU share whatever nproc says
            cores = min(len(os.sched_getaffinity(0)), 16)
            sample = min(frames, 16384 if n_fft <= 1024 else 4096)
            sample -= sample % k_avg
            host = ins[0][:sample].cpu().numpy()
            reps, t_cpu = 0, 0.0
            while t_cpu < 3.0 and reps < 20:       # bounded: a few seconds of CPU work
                c0 = time.perf_counter()
                if cic_only:                   # one thread: the loop carries a dependency
                    if po.ref_available():     # the reference's own object code (oracle/_ref)
                        po.ref_cic_decimate(cic_r, host.reshape(-1, 2))
                    else:
                        po.cic_decimate(cic_r, host.reshape(-1, 2))
                elif cic_r > 1:
                    po.batch_spectra_cic_u8(host, n_fft, cic_r, K=k_avg, nthreads=cores)
                else:
                    po.batch_spectra_u8(host, n_fft, K=k_avg, nthreads=cores)
                t_cpu += time.perf_counter() - c0
                reps += 1
            if cic_only:
                result["cpu_baseline"] = {
                    "value": reps * sample * n_fft / t_cpu, "unit": "samples/s", "cores": 1,
                    "kind": "reference" if po.ref_available() else "port",
                    "sample": "%d x %d decimated outputs of buffer set 0, %d repetitions, cic_decimate of %s"
                              % (sample, n_fft, reps, "the reference's src/resample.c (oracle/_ref)"
                                 if po.ref_available() else "oracle/rtlws_oracle.c")}
            else:
                result["cpu_baseline"] = {
                    "value": reps * sample / t_cpu, "unit": "spectra/s", "cores": cores, "kind": "port",
                    "sample": "%d of the %d frames of buffer set 0, %d repetitions, f64 oracle "
                              "(oracle/rtlws_oracle.c) on %d pthreads" % (sample, frames, reps, cores)}
        print(json.dumps(result), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
