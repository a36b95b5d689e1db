"""bench.py's timed step itself on the GPU (the driver's line carried a
NaN parity block and no test ran run_workload): every default-line workload at a small
frame count must come back with a finite parity block inside its bounds, twice in a row
with identical numbers, and the line must be strict JSON."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built):
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = built.Engine(0)
    yield {"torch": torch, "np": np, "rtlws": built, "eng": eng, "dist": None, "world": 1, "rank": 0,
           "device": dev, "stream": torch.cuda.Stream(device=dev),
           "cu_count": torch.cuda.get_device_properties(dev).multi_processor_count}
    eng.close()


def _small_frames(bench, name):
    n_fft, k_avg, _, _, cic_r, _ = bench.WORKLOADS[name]
    return 512 * k_avg if n_fft * max(cic_r, 1) <= 8192 else 264 * k_avg


@pytest.mark.parametrize("name", ["batched_1024pt_64k_frames", "hann_4096pt_k8_db", "cic8_2048pt", "cic12_2048pt",
                                  "cic10_2048pt", "k6_1024pt_payload", "cic8_block_sums",
                                  "batched_1024pt_64k_frames_f64", "hann_4096pt_k8_db_f64", "rect_2048pt_f64",
                                  "rect_4096pt", "hann_4096pt_k1_db",
                                  "batched_1024pt_64k_frames_f64c_f32o", "hann_4096pt_k8_db_f64c_f32o",
                                  "cic8_2048pt_f64", "cic12_2048pt_f64", "cic8_2048pt_f64c_f32o"])
def test_run_workload_parity_is_finite_bounded_and_repeatable(ctx, name, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "SETTLE_LAUNCHES", 40)        # the ordering matters here, not the governor
    prec = bench.precision_of(name)
    runs = [bench.run_workload(ctx, name, steps=8, warmup=0, sets=4, frames_override=_small_frames(bench, name))
            for _ in range(2)]
    for r in runs:
        par = r["parity"]
        assert "failed" not in par and not par.get("non_finite"), par
        assert bench.parity_failures(par, bench.parity_bounds_for(name)) == []
        json.loads(json.dumps(bench.strict_json(r), allow_nan=False))          # strict JSON
        roof = r["roofline"]
        assert roof["frac_clock"] == "hip_events_on_launch_stream" and roof["frac_wall"] > 0
        assert roof["frac_wall"] <= roof["frac"] * 1.05       # the wall clock contains the events' span
        assert r["dtype"] == ("int32" if name == "cic8_block_sums" else
                              {"f32": "f32", "f64": "f64", "f64c_f32o": "f64 arithmetic, f32 rows"}[prec])
        # the shader clock of this step, between two stamp launches around the leg behind the timed region
        # (rtlws_clock_stamp): nothing is resident beside the launches (round 6: the probe wavefront of rounds 4-5
        # perturbed the timed launches it sat beside)
        assert 0.5 < roof["sclk_ghz"] < 2.6 and "this run" in roof["sclk_source"] and "nothing is resident" in roof["sclk_source"]
        assert roof["sclk_leg_avg_launch_us"] > 0
        if "valu_issue_frac" in roof:                         # (only workloads with a committed instruction count)
            assert "own shader clock" in roof["valu_issue_source"] and 0.0 < roof["valu_issue_frac"] < 1.0   # (small launches: mostly fill and drain)
        n_fft, k_avg, _, output, cic_r, _ = bench.WORKLOADS[name]
        if output in ("power_sum", "mean_db"):               # rows priced at what was stored
            per_frame = 2 * n_fft * max(cic_r, 1) + (8 if prec == "f64" else 4) * n_fft // k_avg
            assert roof["algorithmic_bytes_per_launch"] == per_frame * r["config"]["frames_per_step"]
    assert runs[0]["parity"] == runs[1]["parity"]             # same seed, same kernel: identical statistics


def test_energy_leg_and_box_calibration(ctx, monkeypatch):
    """roofline.energy (the package energy accumulator of THIS device, found by PCI bus id -- ADVICE r5) and
    roofline.box (a fixed v_fma_f64 stream at the cap: the chip's own figure)."""
    import bench
    import rtlws
    monkeypatch.setattr(bench, "SETTLE_LAUNCHES", 40)
    monkeypatch.setattr(bench, "ENERGY_LAUNCHES", 200)
    ec = bench.energy_counter_for(rtlws, 0)
    ctx2 = dict(ctx, energy_counter=ec)
    for name in (bench.HEADLINE, bench.FAST_MODE):
        r = bench.run_workload(ctx2, name, steps=8, warmup=0, sets=2, frames_override=16384)
        assert "failed" not in r["parity"] and bench.parity_failures(r["parity"], bench.parity_bounds_for(name)) == []
        en = r["roofline"].get("energy")
        if en is not None:                    # (rocm_smi readable on the box, and the accumulator moved in these ~6 ms)
            assert ec is not None and en["launches"] == 200 and 0.0 < en["mj_per_launch"] < 1000.0 and 100.0 < en["watts"] < 2000.0
            assert en["bus_id"] == ec.bus_id and en["rsmi_index"] == ec.index
    box = bench.box_calibration(rtlws, 0, ec, settle_s=0.05, measure_s=0.1)
    assert box is not None and 1.0 < box["fma_f64_sclk_ghz_at_cap"] < 2.6
    assert 3.5 < box["cycles_per_instruction_and_simd"] < 6.0          # v_fma_f64: 4 cycles per wave64 instruction
    assert 20.0 < box["fma_f64_tflops"] < 90.0
    if ec is not None:
        assert 300.0 < box["watts"] < 2000.0


def test_run_workload_uniform_input_variant(ctx):
    """SURVEY.md §8d's other input: uniform random bytes (bench.py --input uniform)."""
    import bench
    c = dict(ctx, input="uniform")
    r = bench.run_workload(c, "batched_1024pt_64k_frames", steps=4, warmup=0, sets=2, frames_override=1024)
    assert "uniform" in r["config"]["input"] and "failed" not in r["parity"]


def test_torch_default_stream_handle_is_mapped(built):
    import torch
    assert torch.cuda.current_stream().cuda_stream == 0       # the trap: same bits as "engine's own stream"
    assert built.torch_stream_handle() == built.STREAM_DEFAULT == 1
    side = torch.cuda.Stream()
    assert built.torch_stream_handle(side) == side.cuda_stream != 0


def test_launch_on_hip_default_stream_is_ordered_with_torch(built, oracle):
    """RTLWS_STREAM_DEFAULT: the kernel is enqueued on HIP's default stream, i.e. after the
    torch kernels that produce its input and before the ones that consume its output -- with
    no synchronisation in between."""
    import torch
    from rtlws import synth
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    N, nframes = 1024, 4096
    host = synth.tone_noise_iq(nframes, N, seed=21)
    staged = torch.from_numpy(host).to(dev)
    desc = built.make_desc(N)
    for _ in range(3):
        iq = torch.zeros_like(staged)
        out = torch.full((nframes, N), -1.0, dtype=torch.float32, device=dev)
        big = torch.randn(1 << 26, device=dev)                # keep the default stream busy in front
        big = big * 1.0001 + 1.0
        iq.copy_(staged)                                      # producer, default stream
        eng.spectra_batch(desc, iq.data_ptr(), nframes, out.data_ptr(), stream=built.torch_stream_handle())
        total = out.sum(dtype=torch.float64)                  # consumer, default stream
        got = out.cpu().numpy()
        assert (got >= 0).all()                               # no row still holds the -1 fill
        assert np.isclose(float(total), got.astype(np.float64).sum(), rtol=1e-9)   # the consumer saw the result
        ref = oracle.batch_spectra_u8(host[:64], N)
        rel = np.abs(got[:64] - ref) / np.maximum(ref, 1e-5 * ref.max(axis=1, keepdims=True))
        assert rel.max() <= 1e-4
    eng.close()


def test_two_ranks_rehearsal_on_one_gpu():
    """The N-rank path of bench.py with the real GPU steps, on a box with ONE GPU: two ranks under
    torch.distributed.run share device 0 and rendezvous over gloo (RTLWS_BENCH_REHEARSAL=1; RCCL
    refuses two ranks on one device).  What is checked is the code path the driver's 8-GPU run
    takes -- one line from rank 0, n_gpus = 2, both ranks' own times, the whole-job value = 2 x
    frames x steps / elapsed, parity of what rank 0 timed -- not the numbers."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RTLWS_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "40", "--warmup", "10", "--frames", "8192"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and "rehearsal" in r and "cpu_baseline" not in r and "extra_workloads" not in r
    assert abs(r["value"] - 2 * 40 * 8192 / (r["ms_per_step"] * 40e-3)) / r["value"] < 1e-9
    pr = r["per_rank"]
    assert len(pr["ms_per_step_own"]["all"]) == 2 and len(pr["event_ms_per_step"]["all"]) == 2
    assert pr["ms_per_step_own"]["max"] <= r["ms_per_step"] * 1.001
    assert "failed" not in r["parity"] and r["scaling"] == "weak"


def test_clock_stamps_bracket_a_plausible_clock(ctx):
    """rtlws_clock_stamp: two stamp launches in one stream around a series of launches, paired by place (s_memtime is
    a counter of the place it is read at), give a shader clock in the chip's range over an interval as long as the
    launches, from most of the chip's SIMDs; bad arguments are refused."""
    import torch
    eng, built = ctx["eng"], ctx["rtlws"]
    dev = ctx["device"]
    stream = built.torch_stream_handle()
    iq = torch.randint(0, 256, (4096, 1024, 2), dtype=torch.uint8, device=dev)
    out = torch.empty((4096, 1024), dtype=torch.float32, device=dev)
    desc = built.make_desc(1024)
    slots = 2048
    st = torch.zeros((2, slots, 4), dtype=torch.int64, device=dev)
    eng.clock_stamp(st[0].data_ptr(), slots, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.spectra_batch(desc, iq.data_ptr(), 4096, out.data_ptr(), stream=stream)
    e1.record()
    eng.clock_stamp(st[1].data_ptr(), slots, stream=stream)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    assert (s[:, :, 3] == 0x5354414d50).all()                       # every slot written, both launches
    ghz, secs, places = eng.clock_from_stamps(s[0], s[1])
    assert ghz is not None and 0.5 < ghz < 2.6, ghz
    assert places >= 4 * ctx["cu_count"] // 2                        # at least half of the SIMDs reached by both launches
    assert 0.9 * e0.elapsed_time(e1) * 1e-3 <= secs <= 1.5 * e0.elapsed_time(e1) * 1e-3 + 1e-3
    M = 0x5354414d50
    assert eng.clock_from_stamps([[100, 50, 7, M]], [[300, 50, 7, M]]) == (None, 0.0, 0)            # empty interval
    assert eng.clock_from_stamps([[100, 50, 7, M]], [[300, 150, 8, M]]) == (None, 0.0, 0)           # no common place
    assert eng.clock_from_stamps([[0, 0, 3, M], [5, 0, 4, 0]], [[2000, 100, 3, M]]) == (2.0, 1e-6, 1)
    L = built.hip_lib()
    assert L.rtlws_clock_stamp(eng.h, None, 4, None) == -1
    assert L.rtlws_clock_stamp(eng.h, st[0].data_ptr() + 4, 4, None) == -1 and L.rtlws_clock_stamp(eng.h, st[0].data_ptr(), 0, None) == -1


def test_clock_probe_measures_a_plausible_clock_and_always_leaves(ctx):
    """rtlws_clock_probe_*: the probe wavefront beside a series of launches reports a shader clock in
    the chip's range and the length of the interval; stopping it at once (nothing launched) works too."""
    import time
    eng, rtlws = ctx["eng"], ctx["rtlws"]
    from rtlws import synth
    iq = eng.upload(synth.tone_noise_iq(4096, 1024, seed=3))
    out = eng.alloc(4096 * 1024 * 4)
    desc = rtlws.make_desc(1024)
    probe = eng.clock_probe_start()
    t0 = time.perf_counter()
    for _ in range(300):
        eng.spectra_batch(desc, iq, 4096, out)
    eng.sync()
    dt = time.perf_counter() - t0
    ghz, secs = eng.clock_probe_stop(probe)
    assert 0.5 < ghz < 2.6, ghz
    assert 0.5 * dt < secs < dt + 0.05
    ghz2, secs2 = eng.clock_probe_stop(eng.clock_probe_start())
    assert 0.3 < ghz2 < 2.6 and secs2 < 0.05
