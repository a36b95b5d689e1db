"""Engine option "split" = Q (rtlws_engine_set_option): a batch's rows cut into Q contiguous ranges of
whole K-groups, launched concurrently on engine-owned queues and joined back into the caller's stream
(rtl-ws_amd/csrc/shim.hip: split_launch).  The rows must be the one-launch rows bit for bit -- frames are
independent -- in every arithmetic; the rows are also held against the oracle
(src/spectrum.c:15-35,47-63 restated), so the comparison is never the kernel against itself only.  A
captured and replayed split batch gives the same rows (the side queues fork from and join the capturing
stream)."""
import numpy as np
import pytest

from helpers import rel_err, EPS_K1, EPS_STRICT, TOL, TOL_F64

pytestmark = pytest.mark.gpu


def _run(built, eng, desc, iq_dev, nframes, out, f64):
    fn = eng.spectra_batch_f64 if f64 else eng.spectra_batch
    fn(desc, iq_dev.data_ptr(), nframes, out.data_ptr(), stream=built.torch_stream_handle())


@pytest.mark.parametrize("arith", ["f32", "f64", "f64c_f32o"])
@pytest.mark.parametrize("Q,rows,K", [(2, 9001, 1), (3, 12301, 1), (2, 8200, 2)])
def test_split_rows_identical_to_one_launch(built, oracle, arith, Q, rows, K):
    import torch
    from rtlws import synth
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    N = 1024
    nframes = rows * K
    iq_host = synth.tone_noise_iq(nframes, N, seed=100 + Q + K)
    iq = torch.from_numpy(iq_host).to(dev)
    f64 = arith != "f32"
    desc = built.make_desc(N, k_avg=K, flags=built.FLAG_ROWS_F32 if arith == "f64c_f32o" else 0)
    odt = torch.float64 if arith == "f64" else torch.float32
    one = torch.zeros((rows, N), dtype=odt, device=dev)
    cut = torch.zeros((rows, N), dtype=odt, device=dev)
    assert eng.get_option("split") == 1
    _run(built, eng, desc, iq, nframes, one, f64)
    eng.set_option("split", Q)
    assert eng.get_option("split") == Q
    _run(built, eng, desc, iq, nframes, cut, f64)
    torch.cuda.synchronize()
    assert torch.equal(one, cut)
    # the ranges' seams and ends against the oracle
    bounds = sorted({0, rows - 1} | {r * rows // Q + d for r in range(1, Q) for d in (-1, 0)})
    for r in bounds:
        ref = oracle.batch_spectra_u8(iq_host[r * K:(r + 1) * K], N, K=K)[0]
        got = cut[r].cpu().numpy()
        if arith == "f32":
            assert rel_err(got, ref, EPS_K1 if K == 1 else EPS_STRICT).max() <= TOL
        elif arith == "f64":
            assert rel_err(got, ref, EPS_STRICT).max() <= TOL_F64
        else:
            assert rel_err(got, ref, EPS_STRICT).max() <= 6.0e-8
    # a batch too small to cut is one launch whatever the option says, and still right
    small = torch.zeros((64, N), dtype=odt, device=dev)
    _run(built, eng, built.make_desc(N, flags=desc.flags), iq, 64, small, f64)
    torch.cuda.synchronize()
    ref = oracle.batch_spectra_u8(iq_host[:64], N)
    assert rel_err(small.cpu().numpy(), ref, EPS_K1 if arith == "f32" else EPS_STRICT).max() <= (TOL if arith == "f32" else 6.0e-8)
    eng.close()


def test_split_other_sizes_and_outputs(built, oracle):
    """2048-point CIC-fused payload rows and 4096-point Hann dB rows, cut in two."""
    import torch
    from rtlws import synth
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    for N, K, kw, in_scale in ((2048, 1, dict(cic_r=8, output="payload_u8", gain_db=10), 8), (4096, 2, dict(window="hann", output="mean_db"), 1)):
        rows = 8300
        iq_host = synth.tone_noise_iq(rows * K * in_scale, N, seed=7 + N)
        iq = torch.from_numpy(iq_host).to(dev)
        desc = built.make_desc(N, k_avg=K, **kw)
        odt = torch.uint8 if kw["output"] == "payload_u8" else torch.float32
        one = torch.zeros((rows, N), dtype=odt, device=dev)
        cut = torch.zeros((rows, N), dtype=odt, device=dev)
        eng.set_option("split", 1)
        _run(built, eng, desc, iq, rows * K, one, False)
        eng.set_option("split", 2)
        _run(built, eng, desc, iq, rows * K, cut, False)
        torch.cuda.synchronize()
        assert torch.equal(one, cut)
        assert int(cut.to(torch.int64).sum()) != 0 if odt == torch.uint8 else bool(torch.isfinite(cut).all())
    eng.close()


def test_split_batch_is_capturable(built, oracle):
    import torch
    from rtlws import synth
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    N, rows = 1024, 8192
    assert built.hip_lib().rtlws_engine_prepare(eng.h, N) == 0
    assert built.hip_lib().rtlws_engine_prepare_f64(eng.h, N) == 0
    eng.set_option("split", 2)          # creates the side queue and the events now, outside the capture
    iq_host = synth.tone_noise_iq(rows, N, seed=77)
    iq = torch.from_numpy(iq_host).to(dev)
    out32 = torch.zeros((rows, N), dtype=torch.float32, device=dev)
    out64 = torch.zeros((rows, N), dtype=torch.float32, device=dev)
    d32, d64 = built.make_desc(N), built.make_desc(N, flags=built.FLAG_ROWS_F32)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.spectra_batch(d32, iq.data_ptr(), rows, out32.data_ptr(), stream=built.torch_stream_handle())
            eng.spectra_batch_f64(d64, iq.data_ptr(), rows, out64.data_ptr(), stream=built.torch_stream_handle())
    torch.cuda.current_stream().wait_stream(side)
    assert float(out32.abs().sum()) == 0.0 and float(out64.abs().sum()) == 0.0      # capture enqueued nothing
    g.replay()
    torch.cuda.synchronize()
    ref = oracle.batch_spectra_u8(iq_host, N, nthreads=8)
    assert rel_err(out32.cpu().numpy(), ref, EPS_K1).max() <= TOL
    assert rel_err(out64.cpu().numpy(), ref, EPS_STRICT).max() <= 6.0e-8
    eng.close()
