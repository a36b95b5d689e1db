#!/usr/bin/env python3
"""bench.py -- spectra/s of the fused IQ -> power-spectrum hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch: ONE launch of the fused
kernel over 65 536 frames of 1024 cmplx_u8 (BASELINE.json configs[1]: 128 MiB
of device-resident IQ in, 256 MiB of f32 power spectra out, K=1, rectangular
window = reference behaviour) IN THE REFERENCE'S ARITHMETIC -- double, like
src/spectrum.c:21,28,54-58; each row value rounded once to f32 on the store
(workload batched_1024pt_64k_frames_f64c_f32o; the f32-arithmetic kernel on the
same frames is extra_workloads' "fast mode").  Inputs are resident in HBM before the timed
region; steps rotate over 4 buffer sets (1.5 GiB) so nothing is served from
the 256 MiB Infinity Cache.  The kernel runs at the package power cap, whose
clock governor needs ~25 ms to settle: at least 500 untimed launches precede
the timed region (W of them are the warm-up steps; the rest are reported as
`settle_launches`).

N > 1: one process per GPU, every rank runs the same batch on its own device
(independent frames, no collective on the data path): weak scaling.  Invoked
directly with --gpus N > 1 (no WORLD_SIZE in the environment) this file starts
the N ranks itself -- `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 bench.py ...` as a child process,
before anything touches a GPU -- relays rank 0's JSON line and exits with the
child's code; under torch.distributed.run it is a rank.

One JSON line on stdout (rank 0).  `roofline` prices the kernel against HBM
using the ALGORITHMIC bytes (2*N*R in + 4*N/K out per frame);
`cpu_baseline` times the f64 oracle (oracle/, kind "port") on this host's
cores -- one thread, the job's CPU quota and the whole affinity mask; `value` is
the fastest of them, `cores` the thread count that produced it -- on a bounded
sample of the same workload; `extra_workloads` carries the same measurement (fewer
steps) for BASELINE.json configs[2] and configs[3] -- each in f32 AND in the reference's
arithmetic, each with its own cpu_baseline -- the reference's own CIC factor and the f32
fast mode of configs[1]; `roofline.box` characterises the chip itself (a fixed v_fma_f64
stream at the power cap: the clock and the watts the governor gives it).
"""
import argparse
import json
import os
import socket
import subprocess
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
    # the reference's default rate, 2.048 MS/s (src/rtl_sensor.c:12): 2.048e6 / 192e3 = 10
    "cic10_2048pt": (2048, 1, "rect", "power_sum", 10, 6552),
    # the other fused sizes of the batch API, reference behaviour (rectangular, K = 1)
    "rect_2048pt": (2048, 1, "rect", "power_sum", 0, 32768),
    "rect_4096pt": (4096, 1, "rect", "power_sum", 0, 16384),
    "hann_4096pt_k1_db": (4096, 1, "hann", "mean_db", 0, 16384),          # configs[2] without the averaging
    "rect_4096pt_k8": (4096, 8, "rect", "power_sum", 0, 16384),           # ... without the window
    # the product's own averaging (src/cbb_main.c:18: 6 frames per estimate), byte payload out
    "k6_1024pt_payload": (1024, 6, "rect", "payload_u8", 0, 65536 - 65536 % 6),
    # stand-alone CIC (reference src/resample.c:6-45): "frame" = 2048 decimated outputs,
    # 2*8*2048 bytes in, 8*2048 bytes out; unit reported: decimated samples/s
    "cic8_block_sums": (2048, 1, "rect", "cs32", 8, 8192),
    # the reference's own precision on batches (src/spectrum.c:54-60,28 is f64 end to end):
    # rtlws_spectra_batch_f64, rows of doubles out -- 2N + 8N/K = 10 240 B per spectrum
    "batched_1024pt_64k_frames_f64": (1024, 1, "rect", "power_sum", 0, 65536),
    # configs[2] in f64 (2*4096 + 8*4096/8 = 12 288 B per frame), and the other two fused sizes
    "hann_4096pt_k8_db_f64": (4096, 8, "hann", "mean_db", 0, 16384),
    "rect_2048pt_f64": (2048, 1, "rect", "power_sum", 0, 32768),
    "rect_4096pt_f64": (4096, 1, "rect", "power_sum", 0, 16384),
    # configs[3] in the reference's precision (src/resample.c:21-40 -> src/spectrum.c:65-81, all
    # double): 2*8*2048 + 8*2048 = 49 152 B per spectrum; and the reference's own factor
    "cic8_2048pt_f64": (2048, 1, "rect", "power_sum", 8, 8192),
    "cic12_2048pt_f64": (2048, 1, "rect", "power_sum", 12, 5456),
    # f64 ARITHMETIC, f32 ROWS (RTLWS_FLAG_ROWS_F32): the contract's own byte count -- 6 144 B per
    # 1024-point spectrum, SURVEY.md 8d -- with the reference's arithmetic: the strict metric
    # (floor 1e-9) is one f32 rounding, no builder-chosen floor
    "batched_1024pt_64k_frames_f64c_f32o": (1024, 1, "rect", "power_sum", 0, 65536),
    "hann_4096pt_k8_db_f64c_f32o": (4096, 8, "hann", "mean_db", 0, 16384),
    "cic8_2048pt_f64c_f32o": (2048, 1, "rect", "power_sum", 8, 8192),
}
F64_WORKLOADS = ("batched_1024pt_64k_frames_f64", "hann_4096pt_k8_db_f64", "rect_2048pt_f64", "rect_4096pt_f64",
                 "cic8_2048pt_f64", "cic12_2048pt_f64")
F64C_F32O_WORKLOADS = ("batched_1024pt_64k_frames_f64c_f32o", "hann_4096pt_k8_db_f64c_f32o", "cic8_2048pt_f64c_f32o")
# BASELINE.json configs[1] at the contract's 6 144 B per spectrum in the reference's arithmetic (f64, f32 rows)
HEADLINE = "batched_1024pt_64k_frames_f64c_f32o"
FAST_MODE = "batched_1024pt_64k_frames"          # the same frames in f32 arithmetic: narrower than the reference
# configs[2], configs[3] and the reference's own decimation factor ride along on the default line,
# then configs[1] in f32 (fast mode) and with f64 rows, and configs[3] in the reference's arithmetic
EXTRA_WORKLOADS = ("hann_4096pt_k8_db", "hann_4096pt_k8_db_f64c_f32o", "cic8_2048pt", "cic8_2048pt_f64", "cic12_2048pt",
                   FAST_MODE, "batched_1024pt_64k_frames_f64")
# ... and these carry their own cpu_baseline (the BASELINE.json configurations other than the headline, in f32 and
# in the reference's arithmetic: the CPU path -- the f64 oracle -- is the same for both, so it is timed once per
# configuration and the block shared)
EXTRA_CPU_BASELINE = {"hann_4096pt_k8_db": 0.2, "hann_4096pt_k8_db_f64c_f32o": 0.2,
                      "cic8_2048pt": 0.2, "cic8_2048pt_f64": 0.2}      # name -> budget_scale
EXTRA_STEPS = 200


def precision_of(name):
    """"f32" | "f64" | "f64c_f32o" (f64 arithmetic, rows rounded once to f32)."""
    return "f64" if name in F64_WORKLOADS else "f64c_f32o" if name in F64C_F32O_WORKLOADS else "f32"


def algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output="power_sum", f64=False):
    if output == "cs32":                       # stand-alone CIC: cmplx_s32 out
        return 2 * n_fft * cic_r + 8 * n_fft
    out_bytes = 1 if output == "payload_u8" else (8 if f64 else 4)   # f64 rows are 8N/K
    return 2 * n_fft * max(cic_r, 1) + out_bytes * n_fft // k_avg


def synth_iq_torch(torch, nframes, samples_per_frame, seed, device, kind="tone"):
    """Tone (amp 0.6, random frequency) + Gaussian noise (sigma 0.05), quantised
    to offset-binary u8 -- SURVEY.md §8d's input, generated on the device.
    kind="uniform": §8d's other variant, every byte uniform in 0..255 (the kernels are
    data-independent: this is a parity case, the rate does not move)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((nframes, samples_per_frame, 2), dtype=torch.uint8, device=device)
    if kind == "uniform":
        step = max(1, (1 << 26) // (2 * samples_per_frame))
        for a in range(0, nframes, step):
            b = min(a + step, nframes)
            out[a:b] = torch.randint(0, 256, (b - a, samples_per_frame, 2), generator=g, device=device,
                                     dtype=torch.uint8)
        return out
    chunk = max(1, (1 << 24) // samples_per_frame)
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


# ---- which device a rank uses ------------------------------------------------------

def rehearsal():
    """RTLWS_BENCH_REHEARSAL=1: the N-rank code path on a box with fewer GPUs than ranks -- the
    ranks share devices (rank r on device r mod n) and rendezvous over gloo instead of RCCL (which
    refuses two ranks on one device).  Everything else is the real thing: the GPU steps, the
    barrier-bracketed timing, the MAX over ranks, the per-rank spread, the one JSON line.  The line
    says so and its numbers are not a measurement."""
    return os.environ.get("RTLWS_BENCH_REHEARSAL", "") == "1"


def device_for_rank(local_rank, device_count):
    """One process per GPU: rank r of the node uses device r.  A rank without a device of
    its own is an error, never a silent share -- eight ranks on fewer devices would report
    an 8-GPU rate that is not one.  (Except in a rehearsal, which is labelled as one.)"""
    if device_count < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if rehearsal():
        return local_rank % device_count
    # a launcher may show every rank ONE device of its own through a visibility mask: that device is index 0 in
    # every rank (rank 0 still checks, by PCI bus id, that no two ranks ended up on the same one)
    mask = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
    if device_count == 1 and local_rank > 0 and mask and "," not in mask:
        return 0
    if not 0 <= local_rank < device_count:
        raise SystemExit("bench.py: LOCAL_RANK %d but this host has %d HIP device(s)" % (local_rank, device_count))
    return local_rank


def multi_stream_plan(streams, device_count):
    """BASELINE.json configs[4]'s placement, as rtlws_stream_device_for() states it (include/
    rtlws_stream.h): stream i -> device i mod device_count.  Returns the device of every stream."""
    if streams < 1 or device_count < 1:
        raise ValueError("streams and device_count must be >= 1")
    return [i % device_count for i in range(streams)]


# ---- N > 1 started directly: fan out into one rank per GPU -------------------

def needs_fan_out(gpus, environ):
    """True when this process was asked for N > 1 GPUs but is not itself a rank."""
    return gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def fan_out(gpus, argv, plumbing_cpu=False):
    """Start `gpus` ranks of this file under torch.distributed.run as a CHILD
    process (nothing here has touched a GPU: torch.cuda.device_count() does not
    initialise one on this image), relay rank 0's JSON line, return the child's
    exit code.  Fewer devices than ranks is an error, not a smaller job."""
    if not plumbing_cpu:
        import torch
        have = torch.cuda.device_count()
        if have < gpus:
            print("bench.py: --gpus %d but this host has %d HIP device(s); refusing to report a "
                  "smaller job under that name" % (gpus, have), file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + list(argv)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():
        try:
            if isinstance(json.loads(ln), dict):
                line = ln
        except ValueError:
            sys.stderr.write(ln + "\n")          # anything else a rank wrote to stdout
    if line is not None:
        print(line, flush=True)
    elif child.returncode == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return child.returncode


# ---- one workload: settle, time K steps, price against HBM --------------------

def parity_block(np, po, wl, host_in, got, nchk, f64=False):
    """f64: the rows come from f64 ARITHMETIC (f64 or f32 rows): no relaxed-floor statistics."""
    n_fft, k_avg, window, output, cic_r, _ = wl
    if output == "cs32":
        want = (host_in.astype(np.int32) - 128).reshape(-1, cic_r, 2).sum(axis=1).reshape(nchk, -1)
        return {"frames": nchk, "bit_exact": bool(np.array_equal(got, want))}
    got = got.astype(np.float64)
    if cic_r > 1:
        ref = po.batch_spectra_cic_u8(host_in, n_fft, cic_r, K=k_avg, nthreads=8)
    else:
        ref = po.batch_spectra_u8(host_in, n_fft, K=k_avg, nthreads=8,
                                  window=None if window == "rect" else
                                  (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)))
    if output == "mean_db":
        with np.errstate(divide="ignore"):
            ref_db = 10 * np.log10(ref / k_avg)
        err = np.abs(got - ref_db)
        err[~np.isfinite(ref_db) & ~np.isfinite(got)] = 0.0          # log(0) on both sides (constant frames)
        if k_avg == 1 and not f64:
            # single f32 frames: the budget is stated for bins within 50 dB of the row maximum
            # (1e-4 relative = 4.3e-4 dB); the rest is reported, not bounded
            strong = ref >= 1e-5 * ref.max(axis=1, keepdims=True)
            return {"frames": nchk, "max_abs_db_err_within_50db": float(err[strong].max()),
                    "max_abs_db_err_all_bins": float(err.max())}
        return {"frames": nchk, "max_abs_db_err": float(err.max())}
    if output == "payload_u8":
        want = np.stack([po.spectrum_payload(r, k_avg, 0) for r in ref])
        diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
        return {"frames": nchk, "bytes": int(want.size), "bytes_differing": int((diff != 0).sum()),
                "max_byte_diff": int(diff.max())}
    mx = ref.max(axis=1, keepdims=True)
    # An all-zero oracle row means a CONSTANT input frame (only the DC bin is excited and it
    # is never output, src/spectrum.c:31): the seeded tone + noise never produces one, so
    # the input the kernel consumed is not the input that was synthesised.  Say which rows,
    # and let the caller fail instead of printing 0/0.
    dead = np.nonzero(mx[:, 0] <= 0)[0]
    if dead.size:
        rows_in = host_in.reshape(ref.shape[0], -1)
        return {"frames": nchk, "non_finite": True,
                "constant_input_rows": [int(r) for r in dead[:16]], "constant_input_row_count": int(dead.size),
                "input_byte_min_max_of_those_rows": [[int(rows_in[r].min()), int(rows_in[r].max())] for r in dead[:16]],
                "kernel_output_max_of_those_rows": [float(got[r].max()) for r in dead[:16]]}
    strict = np.abs(got - ref) / np.maximum(ref, 1e-9 * mx)
    return {"frames": nchk,
            "max_rel_err_floor1e-5": float((np.abs(got - ref) / np.maximum(ref, 1e-5 * mx)).max()),
            "max_rel_err_floor1e-9": float(strict.max()),
            "p99.9_rel_err_floor1e-9": float(np.percentile(strict, 99.9))}


# what a parity block must hold for the line to be printed with exit code 0 (the f32 batch
# kernel's budget, DESIGN.md §5: the same bounds tests/ assert)
PARITY_BOUNDS = {"max_rel_err_floor1e-5": 1e-4, "max_rel_err_floor1e-9": 5e-3, "p99.9_rel_err_floor1e-9": 1e-4,
                 "max_abs_db_err": 2e-4, "max_abs_db_err_within_50db": 4.4e-4, "max_byte_diff": 1}
PARITY_BOUNDS_F64 = {"max_rel_err_floor1e-9": 1e-10, "max_abs_db_err": 1e-9}
# f64 arithmetic, f32 rows: one f32 rounding (2^-24 relative; an f32 ulp of a ~100 dB value)
PARITY_BOUNDS_F64C_F32O = {"max_rel_err_floor1e-9": 6.0e-8, "max_rel_err_floor1e-5": 6.0e-8,
                           "p99.9_rel_err_floor1e-9": 6.0e-8, "max_abs_db_err": 1e-5}


def parity_bounds_for(name):
    return {"f64": PARITY_BOUNDS_F64, "f64c_f32o": PARITY_BOUNDS_F64C_F32O}.get(precision_of(name))


def parity_failures(block, bounds=None):
    """Names of the statistics of a parity block that are non-finite or over their bound."""
    import math
    bounds = PARITY_BOUNDS if bounds is None else bounds
    bad = []
    if block.get("non_finite"):
        bad.append("constant_input_rows")
    if block.get("bit_exact") is False:
        bad.append("bit_exact")
    for k, v in block.items():
        if isinstance(v, float) and not math.isfinite(v):
            bad.append(k)
        elif k in bounds and isinstance(v, (int, float)) and v > bounds[k]:
            bad.append(k)
    return bad


def gather_ranks(torch, dist, value, device=None):
    """The value of every rank, in rank order ([value] when dist is None)."""
    if dist is None:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x[0]) for x in out]


def energy_counter_for(rtlws, device_index):
    """The package energy accumulator of HIP device `device_index`, found by its PCI bus id (rocm_smi numbers
    every physical GPU and ignores HIP_VISIBLE_DEVICES: a HIP ordinal is not an rsmi index), or None -- the energy
    leg is then dropped, never read off another GPU.  rtl-ws_amd/rtlws/energy.py."""
    try:
        from rtlws import energy
        return energy.for_hip_device(rtlws, device_index)
    except Exception:
        return None


ENERGY_LAUNCHES = 600          # launches of the leg behind the timed region (energy accumulator, clock stamps)
CLOCK_SLOTS = 2048             # one-wavefront workgroups of a stamp launch: two per SIMD of an MI355X


def run_workload(ctx, name, steps, warmup, sets, frames_override=0, cpu_baseline=False, cpu_budget_scale=1.0):
    """Allocate `sets` rotating buffer sets, run max(warmup, SETTLE_LAUNCHES)
    untimed launches, time exactly `steps` launches between barrier +
    synchronise on both sides (wall clock -> value) and between HIP events on
    the launch stream (-> roofline), MAX over ranks.  Returns rank 0's dict.

    Stream order: inputs are synthesised, and every launch is enqueued, on ONE
    side stream of torch's (ctx["stream"]): its handle is non-zero, so the
    engine launches on exactly that stream (a zero handle -- torch's default
    stream -- would select the engine's own non-blocking stream, which is not
    ordered against torch's kernels: include/rtlws_hip.h "Streams").  The
    outputs are allocated before the synthesis temporaries exist and the
    device is synchronised before the first launch, so no launch can write
    into memory a synthesis kernel still reads."""
    torch, np, rtlws, eng = ctx["torch"], ctx["np"], ctx["rtlws"], ctx["eng"]
    dist, world, rank, device = ctx["dist"], ctx["world"], ctx["rank"], ctx["device"]
    wl = WORKLOADS[name]
    n_fft, k_avg, window, output, cic_r, frames = wl
    prec = precision_of(name)
    f64 = prec != "f32"                    # f64 arithmetic (rtlws_spectra_batch_f64)
    rows64 = prec == "f64"                 # ... with f64 rows
    if frames_override > 0:
        frames = frames_override - frames_override % k_avg
    spf = n_fft * max(cic_r, 1)
    cic_only = (output == "cs32")
    desc = None if cic_only else rtlws.make_desc(n_fft, k_avg, "cu8", window, output, cic_r, 0,
                                                 rtlws.FLAG_ROWS_F32 if prec == "f64c_f32o" else 0)
    rows = frames // k_avg

    # device-resident inputs / outputs, allocated by torch (plumbing only)
    tstream = ctx["stream"]
    stream = tstream.cuda_stream
    assert stream != 0, "bench.py launches on a side stream; a zero handle would select the engine's own"
    out_dtype = torch.int32 if cic_only else (torch.uint8 if output == "payload_u8" else
                                              (torch.float64 if rows64 else torch.float32))
    out_cols = 2 * n_fft if cic_only else n_fft
    with torch.cuda.stream(tstream):
        outs = [torch.empty((rows, out_cols), dtype=out_dtype, device=device) for _ in range(sets)]
        ins = [synth_iq_torch(torch, frames, spf, 1234 + 17 * s + 1000 * rank, device, ctx.get("input", "tone"))
               for s in range(sets)]
    torch.cuda.synchronize()
    L = rtlws.hip_lib()
    launch = eng.spectra_batch_f64 if f64 else eng.spectra_batch

    def step(i):
        s = i % sets
        if cic_only:
            eng.cic_block_sums(cic_r, ins[s].data_ptr(), frames * n_fft, outs[s].data_ptr(), stream=stream)
        else:
            launch(desc, ins[s].data_ptr(), frames, outs[s].data_ptr(), stream=stream)

    # The kernels run at the package power cap and the clock governor needs
    # ~300 launches (25 ms) to settle: whatever W is, at least
    # SETTLE_LAUNCHES untimed launches precede the timed region.
    settle = max(0, SETTLE_LAUNCHES - warmup)
    for i in range(settle + warmup):
        step(i)
    torch.cuda.synchronize()

    ev0, ev1 = L.rtlws_event_create(), L.rtlws_event_create()
    # (a handle creates its HIP event on its first record: do that here, not inside the timed region)
    L.rtlws_event_record(ev0, eng.h, stream)
    L.rtlws_event_record(ev1, eng.h, stream)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # (Nothing but the launches between the two HIP events, and nothing resident beside them: rounds 4-5 kept a
    # clock-probe wavefront on a queue of its own there, which round 6 found to cost the launches +1..7 % -- +38 %
    # for kernels that fill a SIMD's registers; profiles/r06_clock_probe_perturbation.txt.  The shader clock is
    # measured on the leg behind the timed region, below.)
    t0 = time.perf_counter()
    L.rtlws_event_record(ev0, eng.h, stream)
    tA = time.perf_counter()
    for i in range(steps):
        step(i)
    tB = time.perf_counter()
    L.rtlws_event_record(ev1, eng.h, stream)
    tC = time.perf_counter()
    tD = time.perf_counter()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0        # this rank's launches, before waiting for the others
    if os.environ.get("RTLWS_BENCH_DEBUG"):
        print("DEBUG %s: ev0 %.3f ms, enqueue %.3f, ev1 %.3f, stream sync %.3f, device sync %.3f, total %.3f" % (
            name, 1e3 * (tA - t0), 1e3 * (tB - tA), 1e3 * (tC - tB), 1e3 * (tD - tC),
            1e3 * (t0 + own_elapsed - tD), 1e3 * own_elapsed), file=sys.stderr)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = L.rtlws_event_elapsed_ms(ev0, ev1)
    L.rtlws_event_destroy(ev0)
    L.rtlws_event_destroy(ev1)
    # The leg behind the timed region (after the clock has stopped, outside `value`): ENERGY_LAUNCHES more launches
    # of the same step, in the same steady state, with
    #  * the package energy accumulator of this device read before and after (one-GPU jobs) -- joules per launch and
    #    the watts they ran at;
    #  * two stamp launches in the stream around them (include/rtlws_hip.h, rtlws_clock_stamp: CLOCK_SLOTS one-wavefront
    #    workgroups that each write s_memtime, s_memrealtime and the place they ran at, and leave), paired by place:
    #    the shader clock the governor gives this step.  Nothing is resident beside the launches.
    energy, sclk_ghz, probe_s, stamp_places, leg_us = None, None, None, 0, None
    ec = ctx.get("energy_counter") if (world == 1 and ctx.get("energy", True)) else None
    want_clock = ctx.get("clock_probe", True)
    if ec is not None or want_clock:
        stamps = None
        if want_clock:
            try:
                stamps = torch.zeros((2, CLOCK_SLOTS, 4), dtype=torch.int64, device=device)
                eng.clock_stamp(stamps[0].data_ptr(), CLOCK_SLOTS, stream=stream)      # (first use loads the code object)
                torch.cuda.synchronize()
            except Exception as ex:              # an auxiliary measurement: without it the line has no sclk_ghz
                print("bench.py: clock stamp failed (%s); continuing without sclk_ghz" % ex, file=sys.stderr)
                stamps = None
        j0 = ec.joules() if ec is not None else None
        evl0, evl1 = L.rtlws_event_create(), L.rtlws_event_create()
        tq0 = time.perf_counter()
        if stamps is not None:
            L.rtlws_clock_stamp(eng.h, stamps[0].data_ptr(), CLOCK_SLOTS, stream)
        L.rtlws_event_record(evl0, eng.h, stream)
        for i in range(ENERGY_LAUNCHES):
            step(i)
        L.rtlws_event_record(evl1, eng.h, stream)
        if stamps is not None:
            L.rtlws_clock_stamp(eng.h, stamps[1].data_ptr(), CLOCK_SLOTS, stream)
        torch.cuda.synchronize()
        tq1 = time.perf_counter()
        j1 = ec.joules() if ec is not None else None
        leg_us = 1e3 * L.rtlws_event_elapsed_ms(evl0, evl1) / ENERGY_LAUNCHES
        L.rtlws_event_destroy(evl0)
        L.rtlws_event_destroy(evl1)
        if j0 is not None and j1 is not None and j1 > j0:
            energy = {"mj_per_launch": 1e3 * (j1 - j0) / ENERGY_LAUNCHES, "watts": (j1 - j0) / (tq1 - tq0),
                      "launches": ENERGY_LAUNCHES, "avg_launch_us": leg_us, "bus_id": ec.bus_id, "rsmi_index": ec.index,
                      "source": "rocm_smi rsmi_dev_energy_count_get of the GPU at this bus id around %d further "
                                "launches after the timed region (package energy accumulator; not part of "
                                "`value`)" % ENERGY_LAUNCHES}
        if stamps is not None:
            st = stamps.cpu().numpy()
            sclk_ghz, probe_s, stamp_places = eng.clock_from_stamps(st[0], st[1])
    per_rank_sclk = gather_ranks(torch, dist, sclk_ghz or 0.0, ctx.get("reduce_device", device))
    rdev = ctx.get("reduce_device", device)
    elapsed, ev_ms_max = max_over_ranks(torch, dist, [elapsed, ev_ms], rdev)
    per_rank_own = gather_ranks(torch, dist, 1e3 * own_elapsed / steps, rdev)
    per_rank_ev = gather_ranks(torch, dist, ev_ms / steps, rdev)
    # ... and WHERE each rank ran: device, PCI bus id, NUMA node, CPUs it pinned itself to (include/rtlws_topo.h),
    # with its own clock and times -- the first real N-GPU record proves its placement by itself
    ranks = gather_objects(dist, placement_record(rank, ctx.get("local_rank", 0), ctx.get("rank_topology"),
                                                  sclk_ghz=sclk_ghz, ms_per_step_own=1e3 * own_elapsed / steps,
                                                  event_ms_per_step=ev_ms / steps)) if world > 1 else None
    ev_ms = ev_ms_max

    result = None
    if rank == 0:
        value = whole_job_rate(world, steps, frames, elapsed)
        bytes_per_launch = algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output, rows64) * frames
        avg_launch_s = (ev_ms / 1e3) / steps
        achieved = bytes_per_launch / avg_launch_s / 1e9
        achieved_wall = bytes_per_launch / (elapsed / steps) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(name, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
            if traffic is not None:
                traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "workload, committed; NOT measured by this run")
        if cic_only:
            value *= n_fft                      # decimated samples per second
        result = {
            "metric": ("decimated samples/s (CIC R=%d)" % cic_r) if cic_only else
                      ("spectra/s (1024-pt IQ frames)" if n_fft == 1024 else "spectra/s (%d-pt IQ frames)" % n_fft),
            "value": value,
            "unit": "samples/s" if cic_only else "spectra/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "settle_launches": settle,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # arithmetic type of the path (f64c_f32o: f64 arithmetic, rows rounded once to f32)
            "dtype": "int32" if cic_only else {"f32": "f32", "f64": "f64", "f64c_f32o": "f64 arithmetic, f32 rows"}[prec],
            "data": "synthetic",
            "config": {"workload": name, "n_fft": n_fft, "frames_per_step": frames,
                       "k_avg": k_avg, "window": window, "output": output, "cic_r": cic_r,
                       "input": "cmplx_u8 %s, device-resident, %d rotating sets"
                                % ("uniform random bytes" if ctx.get("input") == "uniform" else "tone(0.6)+noise(0.05)", sets),
                       "sharding": "independent frames per GPU, no collective"},
            # frac      : algorithmic bytes / average launch duration between HIP events recorded on the
            #             launch stream around the timed launches (the kernel's own time)
            # frac_wall : the same bytes / ms_per_step, the host wall clock `value` is computed from
            #             (barrier + synchronise on both sides; includes the sync and launch overheads)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_clock": "hip_events_on_launch_stream",
                         "achieved_wall": achieved_wall, "frac_wall": achieved_wall / HBM_PEAK_GBS,
                         "frac_wall_clock": "host_perf_counter_ms_per_step",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": 1e6 * avg_launch_s},
        }
        # valu_issue_frac needs the shader clock of THESE launches (a clock measured on another launch
        # series is not this run's): the probe wavefront above measured it
        if energy is not None:
            result["roofline"]["energy"] = energy
        if sclk_ghz:
            result["roofline"]["sclk_ghz"] = sclk_ghz
            result["roofline"]["sclk_leg_avg_launch_us"] = leg_us
            result["roofline"]["sclk_source"] = (
                "median over %d places (XCC, SE, SH, CU, SIMD) of d(s_memtime) / d(s_memrealtime) x 100 MHz between two stamp "
                "launches in the launch stream around %d further launches of the same step behind the timed region "
                "(%.3f ms, this run); nothing is resident beside the launches" % (stamp_places, ENERGY_LAUNCHES, 1e3 * probe_s))
            vf = valu_issue_frac(name, avg_launch_s, ctx.get("cu_count", 256), sclk_ghz, frames)
            if vf is not None:
                result["roofline"].update(vf)
        if world > 1:
            # every rank's own figures, so a scaling loss is visible in this one line
            result["per_rank"] = {"ranks": ranks,
                                  "ms_per_step_own": {"min": min(per_rank_own), "max": max(per_rank_own),
                                                      "all": per_rank_own},
                                  "event_ms_per_step": {"min": min(per_rank_ev), "max": max(per_rank_ev),
                                                        "all": per_rank_ev},
                                  "sclk_ghz": {"min": min(per_rank_sclk), "max": max(per_rank_sclk), "all": per_rank_sclk},
                                  "note": "own = each rank's wall clock around its launches + synchronise, before "
                                          "the closing barrier; ms_per_step is the MAX over ranks incl. the barrier"}

        # parity spot check of what was just timed (first 256 rows of set 0)
        from oracle import pyoracle as po
        nchk = 256 * k_avg
        host_in = ins[0][:nchk].cpu().numpy()
        got = outs[0][:nchk if cic_only else 256].cpu().numpy()
        result["parity"] = parity_block(np, po, wl, host_in, got, nchk, f64)
        bad = parity_failures(result["parity"], parity_bounds_for(name))
        if bad:
            result["parity"]["failed"] = bad

        if cpu_baseline:
            # the CPU path is the f64 oracle whatever the GPU rows are: one timing per configuration and run
            key = (n_fft, k_avg, window, output, cic_r, frames)
            cache = ctx.setdefault("cpu_baseline_cache", {})
            if key not in cache:
                cache[key] = (name, cpu_baseline_block(np, po, wl, ins[0], frames, cpu_budget_scale))
            first, block = cache[key]
            result["cpu_baseline"] = block if first == name else dict(
                block, shared_with=first, sample=block["sample"] + " [timed once in this run, beside %s: the same "
                "frames through the same f64 CPU path]" % first)
    del ins, outs
    torch.cuda.empty_cache()
    return result


# Vector-issue utilisation of the dominant kernel: the issue cycles one launch needs on a SIMD
# (wave64 VALU instructions per launch, counted by rocprofv3 --pmc SQ_INSTS_VALU, x 2 cycles -- 4 for
# conversions, SDWA and every f64 instruction; tools/make_valu_insts.py, profiles/valu_insts.json) /
# the SIMD-cycles this run's launch lasted (4 SIMDs per CU x CUs x launch time x the shader clock
# rocm-smi showed under sustained load of the same workload, committed beside the count).
def valu_issue_frac(name, avg_launch_s, cu_count, sclk_ghz=None, frames=None):
    """sclk_ghz: the shader clock of the launches `avg_launch_s` was measured on (None: the
    committed clock of another launch series -- only for offline use, never in the bench line).
    frames: frames per launch of this run (None: the committed count's own)."""
    path = os.path.join(ROOT, "profiles", "valu_insts.json")
    try:
        rec = json.load(open(path)).get(name)
    except Exception:
        rec = None
    if not rec:
        return None
    own = sclk_ghz is not None
    simd_cycles = 4 * cu_count * avg_launch_s * (sclk_ghz if own else rec["sclk_ghz_under_load"]) * 1e9
    issue = rec["issue_cycles_per_launch"] * (1.0 if frames is None else frames / rec["frames_per_launch"])
    return {"valu_issue_frac": issue / simd_cycles,
            "valu_issue_source": ("profiles/valu_insts.json (SQ_INSTS_VALU per launch by issue cost: committed count) / "
                                  "this run's launch duration at this run's own shader clock (%.3f GHz)" % sclk_ghz) if own else
                                 "profiles/valu_insts.json (SQ_INSTS_VALU per launch by issue cost, sclk under "
                                 "load: committed measurements of ANOTHER launch series) / this run's launch duration"}


def cpu_quota(cgroup_root="/sys/fs/cgroup", proc_cgroup="/proc/self/cgroup"):
    """CPUs the job's cgroup lets it run at once -- quota / period of cpu.max (cgroup v2) or of
    cpu.cfs_quota_us / cpu.cfs_period_us (v1), the smallest over this process's cgroup and its
    ancestors under `cgroup_root` -- or None when nothing limits it (or nothing can be read)."""
    def read(path):
        try:
            with open(path) as f:
                return f.read().split()
        except OSError:
            return None

    rel = []
    try:
        with open(proc_cgroup) as f:
            lines = f.read().splitlines()
    except OSError:
        lines = []
    for ln in lines:
        parts = ln.split(":", 2)
        if len(parts) == 3 and (parts[1] == "" or "cpu" in parts[1].split(",")):
            rel.append((parts[1], parts[2].strip("/")))
    best = None
    dirs = set()
    for ctrl, path in rel or [("", "")]:
        for base in (cgroup_root, os.path.join(cgroup_root, "cpu"), os.path.join(cgroup_root, "cpu,cpuacct")):
            comps = [c for c in path.split("/") if c]
            for k in range(len(comps) + 1):
                dirs.add(os.path.join(base, *comps[:k]))
    for d in sorted(dirs):
        v2 = read(os.path.join(d, "cpu.max"))
        if v2 and len(v2) == 2 and v2[0] != "max":
            q = float(v2[0]) / float(v2[1])
            best = q if best is None else min(best, q)
        q1, p1 = read(os.path.join(d, "cpu.cfs_quota_us")), read(os.path.join(d, "cpu.cfs_period_us"))
        if q1 and p1 and float(q1[0]) > 0:
            q = float(q1[0]) / float(p1[0])
            best = q if best is None else min(best, q)
    return best


def cpu_thread_counts(mask_cores, quota):
    """The thread counts the CPU path is timed with: one, the job's CPU quota (whole CPUs, at least
    one, never more than the affinity mask) when a quota exists and is smaller than the mask, and the
    whole mask.  Ascending, no duplicates."""
    import math
    counts = {1, mask_cores}
    if quota is not None:
        counts.add(max(1, min(mask_cores, int(math.ceil(quota - 1e-9)))))
    return sorted(counts)


def cpu_baseline_block(np, po, wl, dev_in, frames, budget_scale=1.0, quota="read"):
    """The f64 oracle on this host's cores over a bounded sample of buffer set 0: one
    thread for 2 s, then the job's cgroup CPU quota and the whole affinity mask for 1 s / 0.75 s each --
    about 30 CPU-seconds in all.  `value` is the FASTEST of them and `cores` the thread count that
    produced it (a baseline that understates the CPU would flatter the GPU); every run is in `runs`."""
    n_fft, k_avg, window, output, cic_r, _ = wl
    nproc = os.cpu_count()
    # every core this process may run on (its affinity mask) and what its cgroup lets it use at once
    mask = len(os.sched_getaffinity(0))
    if quota == "read":
        quota = cpu_quota()
    counts = cpu_thread_counts(mask, quota)
    base = 16384 if n_fft <= 1024 else 4096
    # enough rows that every thread has >= 64 frames between its creation and its join
    sample = min(frames, max(base, 64 * mask))
    sample -= sample % k_avg
    host = dev_in[:sample].cpu().numpy()
    win = None if window == "rect" else (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft))

    def timed(nthreads, nframes, budget_s, max_reps=100000):
        h = host[:nframes]
        out = np.zeros((nframes // k_avg, n_fft), dtype=np.float64)    # touched once, reused
        reps, t_cpu = 0, 0.0
        while t_cpu < budget_s and reps < max_reps:
            c0 = time.perf_counter()
            if cic_r > 1:
                po.batch_spectra_cic_u8(h, n_fft, cic_r, K=k_avg, window=win, nthreads=nthreads, out=out)
            else:
                po.batch_spectra_u8(h, n_fft, K=k_avg, window=win, nthreads=nthreads, out=out)
            if output == "mean_db":                # the workload's epilogue belongs to the CPU path too
                with np.errstate(divide="ignore"):
                    np.log10(out / k_avg)
            t_cpu += time.perf_counter() - c0
            reps += 1
        return reps * nframes / t_cpu, reps

    if output == "cs32":                       # one thread: the loop carries a dependency
        reps, t_cpu = 0, 0.0
        while t_cpu < 3.0 * budget_scale and reps < 20:
            c0 = time.perf_counter()
            if po.ref_available():             # the reference's own object code (oracle/_ref)
                po.ref_cic_decimate(cic_r, host.reshape(-1, 2))
            else:
                po.cic_decimate(cic_r, host.reshape(-1, 2))
            t_cpu += time.perf_counter() - c0
            reps += 1
        return {"value": reps * sample * n_fft / t_cpu, "unit": "samples/s", "cores": 1, "nproc": nproc,
                "kind": "reference" if po.ref_available() else "port",
                "sample": "%d x %d decimated outputs of buffer set 0, %d repetitions, cic_decimate of %s"
                          % (sample, n_fft, reps, "the reference's src/resample.c (oracle/_ref)"
                             if po.ref_available() else "oracle/rtlws_oracle.c")}
    one_n = max(k_avg, (min(sample, base) // 8) - (min(sample, base) // 8) % k_avg)
    runs = []
    for n in counts:
        if n == 1:
            nframes, budget = one_n, 2.0
        elif n == mask and len(counts) == 3:
            nframes, budget = sample, 0.75           # the whole mask beside a smaller quota: throttled, short
        else:
            nframes, budget = (sample if n == mask else min(sample, max(base, 64 * n)) - min(sample, max(base, 64 * n)) % k_avg), 1.0
        v, reps = timed(n, nframes, budget * budget_scale)
        runs.append({"value": v, "unit": "spectra/s", "cores": n,
                     "sample": "%d frames of buffer set 0, %d repetitions" % (nframes, reps)})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "spectra/s", "cores": best["cores"], "kind": "port", "nproc": nproc,
            "affinity_mask_cores": mask, "cgroup_cpu_quota": quota,
            "one_thread": runs[0], "runs": runs,
            "sample": "%s (%d-point%s%s, K = %d) of the %d frames of buffer set 0, f64 oracle (oracle/rtlws_oracle.c) on "
                      "%d pthreads -- the fastest of %s threads (one / the cgroup CPU quota / the affinity mask)"
                      % (best["sample"], n_fft, ", CIC %d:1 first" % cic_r if cic_r > 1 else "",
                         ", Hann, mean dB" if win is not None else "", k_avg, frames, best["cores"],
                         " / ".join(str(r["cores"]) for r in runs))}


REALTIME_WORKLOAD = "realtime_8x2400k"


def realtime_main(args):
    """--workload realtime_8x2400k: BASELINE.json configs[4] in its real-time form -- eight paced
    2.4 MS/s sensors (131 072-sample buffers, src/signal_source.c:29-35), stream i on device
    i mod n_devices, host-fed over PCIe -- through rtl-ws_amd/lib/rtlws_multi_stream (C, one
    producer thread and one rtlws_stream per sensor).  Not a roofline workload: what is
    reported is whether every stream keeps up (drops, latency), per stream and per device."""
    import rtlws
    exe = os.path.join(rtlws.LIB_DIR, "rtlws_multi_stream")
    if not os.path.exists(exe):
        rtlws.build()
    ndev = rtlws.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    use = min(ndev, args.gpus) if args.gpus > 1 else ndev
    seconds = max(1.0, min(20.0, args.steps / 1000.0))
    cmd = [exe, "--streams", "8", "--seconds", "%.2f" % seconds, "--rate", "2400000", "--output", "payload",
           "--devices", str(use)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120 + 2 * seconds)
    if out.returncode != 0:
        sys.stderr.write(out.stderr)
        return out.returncode
    r = json.loads(out.stdout.strip().splitlines()[-1])
    per_dev = {}
    for st in r["per_stream"]:
        d = per_dev.setdefault(st["device"], {"device": st["device"], "streams": [], "spectra_per_s": 0.0,
                                              "chunks_dropped": 0, "chunks_failed": 0, "latency_ms_max": 0.0})
        d["streams"].append(st["stream"])
        d["spectra_per_s"] += st["spectra_per_s"]
        d["chunks_dropped"] += st["chunks_dropped"]
        d["chunks_failed"] += st["chunks_failed"]
        d["latency_ms_max"] = max(d["latency_ms_max"], st["latency_ms_max"])
    want = multi_stream_plan(8, r["devices"])
    line = {"metric": "spectra/s (8 paced 2.4 MS/s IQ streams, 1024-pt, host-fed)", "value": r["spectra_per_s_total"],
            "unit": "spectra/s", "n_gpus": r["devices"], "steps": args.steps, "warmup": 0,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": REALTIME_WORKLOAD, "streams": 8, "rate_hz": 2400000, "n_fft": 1024,
                       "output": "payload_u8", "seconds": r["seconds"],
                       "sharding": "stream i on device i mod n_devices, no collective",
                       "real_time_rate_per_stream": 2400000 / 1024.0},
            "roofline": None,
            "realtime": {"chunks_dropped": r["chunks_dropped"], "chunks_failed": r["chunks_failed"],
                         "latency_ms_avg": r["latency_ms_avg"], "latency_ms_max": r["latency_ms_max"],
                         "stream_devices": [st["device"] for st in r["per_stream"]],
                         "placement_as_specified": [st["device"] for st in r["per_stream"]] == want,
                         "per_device": [per_dev[k] for k in sorted(per_dev)], "per_stream": r["per_stream"]}}
    print(json.dumps(strict_json(line), allow_nan=False), flush=True)
    return 0 if (r["chunks_dropped"] == 0 and r["chunks_failed"] == 0) else 5


MULTI_BATCH_WORKLOAD = "multi_batch"


def multi_batch_main(args):
    """--workload multi_batch: BASELINE.json configs[1] sharded over the devices of the node by the
    C host itself (rtl-ws_amd/lib/rtlws_multi_batch over include/rtlws_multi.h: one batch of
    65 536 x G frames, shard g = rows [g*R/G, (g+1)*R/G) on device g, one host pthread + one engine
    per device, per-device HIP-event times, no collective) -- the same measurement as the default
    line without torch.distributed in the way.  --gpus N uses the first N devices (default: all)."""
    import rtlws
    exe = os.path.join(rtlws.LIB_DIR, "rtlws_multi_batch")
    if not os.path.exists(exe):
        rtlws.build()
    ndev = rtlws.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    use = args.gpus if args.gpus > 1 else ndev
    if use > ndev:
        print("bench.py: --gpus %d but this host has %d HIP device(s)" % (use, ndev), file=sys.stderr)
        return 2
    per_gpu = args.frames if args.frames > 0 else WORKLOADS[HEADLINE][5]
    cmd = [exe, "--frames", str(per_gpu * use), "--launches", str(args.steps), "--warmup", str(max(args.warmup, SETTLE_LAUNCHES)),
           "--devices", str(use), "--shards-per-device", str(args.shards_per_device)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        sys.stderr.write(out.stderr)
        return out.returncode
    r = json.loads(out.stdout.strip().splitlines()[-1])
    # per device: the bytes its shards moved in the job's wall time (with one shard per device that is the
    # shard's own HIP-event time; with several, their launches overlap and only the aggregate is a rate)
    q = args.shards_per_device
    if q == 1:
        worst = max(s["event_ms_per_launch"] for s in r["per_shard"])
    else:
        worst = r["wall_ms"] / args.steps
    bytes_per_dev = r["algorithmic_bytes_per_frame"] * per_gpu
    achieved = bytes_per_dev / (worst * 1e-3) / 1e9
    line = {"metric": "spectra/s (1024-pt IQ frames)", "value": r["spectra_per_s_total"], "unit": "spectra/s",
            "n_gpus": r["shards"] // max(1, args.shards_per_device), "steps": args.steps, "warmup": r["warmup"],
            "ms_per_step": r["wall_ms"] / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": MULTI_BATCH_WORKLOAD, "n_fft": r["n_fft"], "k_avg": r["k_avg"],
                       "frames_per_step": r["frames_used"], "frames_per_gpu": per_gpu, "shards_per_device": q,
                       "sharding": "C host: shard g = rows [g*R/G, (g+1)*R/G) on device g, one pthread + one engine "
                                   "per device, no collective (include/rtlws_multi.h)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_clock": "hip_events_per_device_slowest_shard" if q == 1 else
                                       "wall_clock_of_the_job_per_device (concurrent shards overlap: not one kernel's duration)",
                         "traffic": None, "algorithmic_bytes_per_launch": bytes_per_dev,
                         "avg_launch_us": 1e3 * worst},
            "per_device": r["per_shard"]}
    print(json.dumps(strict_json(line), allow_nan=False), flush=True)
    return 0


def plumbing_main(args):
    """--plumbing-cpu (tests only): the rank plumbing of this file with the GPU step
    replaced by a sleep -- rendezvous over gloo, barrier-bracketed timing, MAX over
    ranks, whole-job rate, one JSON line from rank 0.  Measures nothing."""
    import torch
    dist, world, rank = init_distributed(torch, "gloo")
    frames = WORKLOADS[args.workload][5]
    device_for_rank(int(os.environ.get("LOCAL_RANK", "0")), max(world, 1))     # the same rule, CPU ranks
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    own = time.perf_counter() - t0                 # this rank's steps, before waiting for the others
    if dist is not None:
        dist.barrier()
    elapsed, slowest = max_over_ranks(torch, dist, [time.perf_counter() - t0, float(rank)])
    per_rank_own = gather_ranks(torch, dist, 1e3 * own / args.steps)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    topo = None
    if world > 1:
        import rtlws
        topo = pin_rank_to_its_gpu(rtlws, local_rank, planned_topology(rtlws, local_rank) or (lambda _d: None))
    ranks = gather_objects(dist, placement_record(rank, local_rank, topo, sclk_ghz=None,
                                                  ms_per_step_own=1e3 * own / args.steps, event_ms_per_step=None))
    clashes = check_distinct_devices(ranks) if rank == 0 else []
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no GPU step)", "value": whole_job_rate(world, args.steps, frames, elapsed),
                          "n_gpus": world, "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps,
                          "slowest_rank": slowest, "asked_gpus": args.gpus, "data": "none",
                          "per_rank": {"ranks": ranks, "device_clashes": clashes,
                                       "ms_per_step_own": {"min": min(per_rank_own), "max": max(per_rank_own),
                                                           "all": per_rank_own}}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if world != args.gpus else 6 if clashes else 0


def cpus_of_cpulist(text):
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.replace(" ", "").split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def pin_rank_to_its_gpu(rtlws, device, describe=None):
    """Restrict this process to the CPUs of `device`'s NUMA node that its mask already allows; returns what was
    learnt and done ({"numa_node", "bus_id", "cpus_pinned"}).  Nothing known, or nothing of the node inside the
    mask: the process keeps its mask (cpus_pinned 0)."""
    t = (describe or rtlws.topo_describe)(device)
    info = {"numa_node": -1, "bus_id": "", "cpus_pinned": 0}
    if t is None:
        return info
    info["numa_node"], info["bus_id"] = int(t.numa_node), t.bus_id.decode()
    want = set(cpus_of_cpulist(t.cpulist.decode())) & os.sched_getaffinity(0)
    if want:
        os.sched_setaffinity(0, want)
        info["cpus_pinned"] = len(want)
    return info


def planned_topology(rtlws, local_rank):
    """--plumbing-cpu and offline planning: RTLWS_BENCH_BUS_IDS (comma list, rank r takes entry r) and
    RTLWS_BENCH_SYSFS_ROOT describe the rank's device without touching a GPU (include/rtlws_topo.h takes a sysfs
    root for exactly that); None when they are not set."""
    ids = [b for b in os.environ.get("RTLWS_BENCH_BUS_IDS", "").split(",") if b]
    if not ids:
        return None
    bus = ids[local_rank % len(ids)]
    return lambda _device: rtlws.topo_describe(bus_id=bus, sysfs_root=os.environ.get("RTLWS_BENCH_SYSFS_ROOT") or None)


def gather_objects(dist, obj):
    """obj of every rank, in rank order ([obj] when dist is None)."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def placement_record(rank, local_rank, topo, **figures):
    """What one rank contributes to per_rank.ranks: where it ran and its own figures."""
    rec = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(),
           "device": local_rank if topo is None else topo.get("device", local_rank),
           "bus_id": (topo or {}).get("bus_id", ""), "numa_node": (topo or {}).get("numa_node", -1),
           "cpus_pinned": (topo or {}).get("cpus_pinned", 0),
           # a launcher may give every rank ONE visible device (then every rank says "device 0"): the mask tells
           # such ranks apart where the bus id is unknown
           "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""}
    rec.update(figures)
    return rec


def check_distinct_devices(records, allow_shared=False):
    """One process per GPU: no two ranks of a host on one device (bus id where known, else host + device index
    under the rank's visibility mask).  Returns the list of clashes; rank 0 refuses to print an N-GPU line over them
    unless `allow_shared` (rehearsal)."""
    seen, clashes = {}, []
    for r in records:
        key = (r["host"], r["bus_id"] or "device %d of [%s]" % (r["device"], r.get("visible_devices", "")))
        if key in seen:
            clashes.append({"ranks": [seen[key], r["rank"]], "host": key[0], "device": key[1]})
        seen.setdefault(key, r["rank"])
    return [] if allow_shared else clashes


def box_calibration(rtlws, device_index, ec, settle_s=0.15, measure_s=0.25):
    """roofline.box: what THIS chip gives a fixed, memory-free instruction stream at the package power cap
    (rtl-ws_amd/bench/box_calib.hip: v_fma_f64 on every SIMD, two wavefronts each) -- the shader clock the governor
    settles at, the watts (package energy accumulator around the measured launches) and the f64 rate.  The product
    kernels run at the same cap, and what the cap buys differs from chip to chip (0.443-0.487 of the HBM roofline
    for the same code): this separates a slow box from a regression.  After the timed region, never in `value`."""
    import ctypes as C

    class Res(C.Structure):
        _fields_ = [("sclk_ghz", C.c_double), ("gpu_seconds", C.c_double), ("wall_seconds", C.c_double),
                    ("wave_instructions", C.c_double), ("launches", C.c_long), ("simds", C.c_int)]

    path = os.path.join(rtlws.LIB_DIR, "librtlws_bench.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.rtlws_box_calib_run.argtypes = [C.c_int, C.c_double, C.POINTER(Res)]
    r = Res()
    if lib.rtlws_box_calib_run(device_index, settle_s, C.byref(r)) != 0:       # the governor's transient, unmeasured
        return None
    j0 = ec.joules() if ec is not None else None
    if lib.rtlws_box_calib_run(device_index, measure_s, C.byref(r)) != 0:
        return None
    j1 = ec.joules() if ec is not None else None
    box = {"fma_f64_sclk_ghz_at_cap": r.sclk_ghz,
           "fma_f64_tflops": 2.0 * 64.0 * r.wave_instructions / r.gpu_seconds / 1e12,
           "cycles_per_instruction_and_simd": r.sclk_ghz * 1e9 * r.gpu_seconds * r.simds / r.wave_instructions,
           "seconds": r.gpu_seconds, "launches": int(r.launches),
           "kernel": "box_calib_kernel: 256 x v_fma_f64 per loop iteration on eight register pairs, 2 wavefronts per "
                     "SIMD on every CU, no memory traffic (rtl-ws_amd/bench/box_calib.hip); %.2f s settle + this"
                     % settle_s}
    if j0 is not None and j1 is not None and j1 > j0:
        box["watts"] = (j1 - j0) / r.wall_seconds
    return box


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500,
                    help="untimed launches first; the clock governor needs ~300 (25 ms) to settle at the power cap")
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS) + [REALTIME_WORKLOAD, MULTI_BATCH_WORKLOAD])
    ap.add_argument("--sets", type=int, default=4, help="rotating buffer sets")
    ap.add_argument("--input", default="tone", choices=["tone", "uniform"],
                    help="synthetic IQ: tone + noise (SURVEY.md 8d, default) or uniform random bytes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads on the default line")
    ap.add_argument("--frames", type=int, default=0, help="override frames per step (experiments)")
    ap.add_argument("--shards-per-device", type=int, default=1, choices=[1, 2, 3, 4],
                    help="--workload multi_batch: concurrent shards (own queue and host thread) per device")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration (roofline.box)")
    ap.add_argument("--no-energy", action="store_true", help="skip the energy leg after the timed region")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="no clock stamps around the leg behind the timed region (with --no-energy: no such leg at all)")
    ap.add_argument("--plumbing-cpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    if args.workload == REALTIME_WORKLOAD:          # one process drives every device (threads, no ranks)
        return realtime_main(args)
    if args.workload == MULTI_BATCH_WORKLOAD:       # ... and so does the C batch driver
        return multi_batch_main(args)
    if needs_fan_out(args.gpus, os.environ):
        return fan_out(args.gpus, argv, args.plumbing_cpu)
    if args.plumbing_cpu:
        return plumbing_main(args)

    import numpy as np
    import torch
    import rtlws

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    local_rank = device_for_rank(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist, world, rank = init_distributed(torch, "gloo" if rehearsal() else "nccl", device)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # One process per GPU: under N > 1 each rank keeps to the CPUs of its own GPU's NUMA node (include/rtlws_topo.h),
    # so that on a two-socket node no rank's launches, events and barriers cross the sockets.  A one-GPU job keeps
    # its whole mask: its cpu_baseline leg is meant to see the job's cores.
    rank_topology = pin_rank_to_its_gpu(rtlws, local_rank, planned_topology(rtlws, local_rank)) if world > 1 else None

    eng = rtlws.Engine(local_rank)
    ctx = {"torch": torch, "np": np, "rtlws": rtlws, "eng": eng, "dist": dist, "world": world,
           "rank": rank, "device": device,
           # ONE side stream carries input synthesis and every launch (run_workload, "Stream order")
           "input": args.input, "clock_probe": not args.no_clock_probe, "energy": not args.no_energy,
           "stream": torch.cuda.Stream(device=device),
           "cu_count": torch.cuda.get_device_properties(device).multi_processor_count,
           "local_rank": local_rank, "rank_topology": rank_topology,
           # the package energy accumulator of THIS device, by PCI bus id (one-GPU jobs)
           "energy_counter": energy_counter_for(rtlws, local_rank) if (world == 1 and not args.no_energy) else None}

    # First collective = RCCL's lazy communicator set-up (~16 ms): do it here, not
    # between the warm-up launches and the timed region, where that much idle
    # time would put the timed launches back into the governor's transient.
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()

    result = run_workload(ctx, args.workload, args.steps, args.warmup, args.sets, args.frames,
                          cpu_baseline=(world == 1 and not args.no_cpu_baseline))

    # what this chip gives a fixed instruction stream at the cap (roofline.box): right behind the headline's energy
    # leg, the package still at its cap
    if world == 1 and rank == 0 and not args.no_box and not args.no_energy and ctx["clock_probe"]:
        try:
            box = box_calibration(rtlws, local_rank, ctx["energy_counter"])
        except Exception as ex:            # an auxiliary measurement: the line goes out without it
            print("bench.py: box calibration failed (%s); continuing without roofline.box" % ex, file=sys.stderr)
            box = None
        if box is not None:
            result["roofline"]["box"] = box

    # The default line also carries configs[2], configs[3] and the reference's own CIC
    # factor, measured the same way with fewer steps (rank 0 of a 1-GPU job only).
    if world == 1 and args.workload == HEADLINE and not args.no_extra and args.frames == 0:
        extras = []
        for name in EXTRA_WORKLOADS:
            with_cpu = name in EXTRA_CPU_BASELINE and not args.no_cpu_baseline
            r = run_workload(ctx, name, EXTRA_STEPS, 0, args.sets, cpu_baseline=with_cpu,
                             cpu_budget_scale=EXTRA_CPU_BASELINE.get(name, 1.0))
            x = {"workload": name, "metric": r["metric"], "value": r["value"], "unit": r["unit"],
                 "dtype": r["dtype"], "steps": r["steps"], "settle_launches": r["settle_launches"],
                 "ms_per_step": r["ms_per_step"], "config": r["config"],
                 "roofline": r["roofline"], "parity": r["parity"]}
            if "cpu_baseline" in r:
                x["cpu_baseline"] = r["cpu_baseline"]
            extras.append(x)
        result["extra_workloads"] = extras
        # the same frames and the same 6 144 B per spectrum in f32 arithmetic -- narrower than the reference
        # (src/spectrum.c is double end to end), and <= 1e-4 only under the relaxed floor its parity block
        # names: context beside the headline, not the metric
        for x in extras:
            if x["workload"] == FAST_MODE:
                result["fast_mode_line"] = {
                    "workload": x["workload"], "dtype": x["dtype"], "value": x["value"], "unit": x["unit"],
                    "roofline_frac": x["roofline"]["frac"],
                    "max_rel_err_floor1e-9": x["parity"].get("max_rel_err_floor1e-9"),
                    "max_rel_err_floor1e-5": x["parity"].get("max_rel_err_floor1e-5"),
                    "algorithmic_bytes_per_spectrum": 6144}

    rc = 0
    if rank == 0 and world > 1:
        clashes = check_distinct_devices(result["per_rank"]["ranks"], allow_shared=rehearsal())
        if clashes:
            result["per_rank"]["device_clashes"] = clashes
            print("bench.py: ranks share a device: %s -- not an N-GPU measurement" % json.dumps(clashes), file=sys.stderr)
            rc = 6
    if rank == 0 and rehearsal():
        result["rehearsal"] = ("RTLWS_BENCH_REHEARSAL=1: %d ranks share %d device(s) over gloo -- the N-rank code path, "
                               "not a measurement" % (world, torch.cuda.device_count()))
    if rank == 0:
        # A parity block that is non-finite or over its bound is a FAILED run: the line is still
        # printed (strict JSON: a non-finite number becomes a string), the exit code says so.
        blocks = [(result["config"]["workload"], result["parity"])]
        blocks += [(x["workload"], x["parity"]) for x in result.get("extra_workloads", [])]
        failed = {n: b["failed"] for n, b in blocks if b.get("failed")}
        if failed:
            result["parity_failed"] = failed
            print("bench.py: PARITY FAILED on %s -- the timed launches did not reproduce the oracle"
                  % json.dumps(failed), file=sys.stderr)
            rc = 4
        print(json.dumps(strict_json(result), allow_nan=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def strict_json(x):
    """NaN / inf have no JSON spelling: replace them by strings, recursively."""
    import math
    if isinstance(x, float) and not math.isfinite(x):
        return repr(x)
    if isinstance(x, dict):
        return {k: strict_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [strict_json(v) for v in x]
    return x


if __name__ == "__main__":
    sys.exit(main())
