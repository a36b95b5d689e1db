"""The batch launch is capturable into a hipGraph once the tables exist
(rtlws_engine_prepare): ten launches captured with torch.cuda.graph on a side
stream, replayed twice, results equal to eager launches and to the oracle."""
import numpy as np
import pytest

from helpers import rel_err, EPS_K1, TOL

pytestmark = pytest.mark.gpu


def test_capture_and_replay(built, oracle):
    import torch
    from rtlws import synth
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    N, nframes, launches = 1024, 512, 10
    assert built.hip_lib().rtlws_engine_prepare(eng.h, N) == 0
    assert built.hip_lib().rtlws_engine_prepare(eng.h, 1) == -1
    iq_host = synth.tone_noise_iq(nframes * launches, N, seed=9).reshape(launches, nframes, N, 2)
    iq = torch.from_numpy(iq_host).to(dev)
    out = torch.zeros((launches, nframes, N), dtype=torch.float32, device=dev)
    desc = built.make_desc(N)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(launches):
                eng.spectra_batch(desc, iq[i].data_ptr(), nframes, out[i].data_ptr(),
                                  stream=built.torch_stream_handle())
    torch.cuda.current_stream().wait_stream(side)
    assert float(out.abs().sum()) == 0.0            # capture enqueued nothing
    g.replay()
    torch.cuda.synchronize()
    first = out.clone()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, first)                  # replays are bit-identical
    ref = oracle.batch_spectra_u8(iq_host[3], N, nthreads=8)
    assert rel_err(out[3].cpu().numpy(), ref, EPS_K1).max() <= TOL
    eager = torch.zeros((nframes, N), dtype=torch.float32, device=dev)
    eng.spectra_batch(desc, iq[3].data_ptr(), nframes, eager.data_ptr(),
                      stream=built.torch_stream_handle())
    torch.cuda.synchronize()
    assert torch.equal(eager, out[3])
    eng.close()


@pytest.mark.parametrize("N,nframes", [(1024, 256), (1024, 8192), (4096, 64)])
def test_capture_and_replay_f64(built, oracle, N, nframes):
    """The same for rtlws_spectra_batch_f64 once rtlws_engine_prepare_f64 has built its tables:
    f64 rows, and f64 arithmetic with f32 rows, captured and replayed.  8 192 frames of 1024 points take
    the eight-wavefront workgroups (136 KiB of LDS), 4096-point frames need 69.6 KiB: both above the
    64 KiB a kernel gets without hipFuncSetAttribute, which rtlws_engine_prepare_f64 has already called --
    the FIRST launch of each instantiation happens inside the capture here."""
    import torch
    from rtlws import synth
    from helpers import EPS_STRICT
    dev = torch.device("cuda", 0)
    eng = built.Engine(0)
    launches = 4 if nframes <= 256 else 1
    assert built.hip_lib().rtlws_engine_prepare_f64(eng.h, N) == 0
    assert built.hip_lib().rtlws_engine_prepare_f64(eng.h, 9000) == -1
    iq_host = synth.tone_noise_iq(nframes * launches, N, seed=19).reshape(launches, nframes, N, 2)
    iq = torch.from_numpy(iq_host).to(dev)
    out64 = torch.zeros((launches, nframes, N), dtype=torch.float64, device=dev)
    out32 = torch.zeros((launches, nframes, N), dtype=torch.float32, device=dev)
    d64, d32 = built.make_desc(N), built.make_desc(N, flags=built.FLAG_ROWS_F32)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(launches):
                eng.spectra_batch_f64(d64, iq[i].data_ptr(), nframes, out64[i].data_ptr(), stream=built.torch_stream_handle())
                eng.spectra_batch_f64(d32, iq[i].data_ptr(), nframes, out32[i].data_ptr(), stream=built.torch_stream_handle())
    torch.cuda.current_stream().wait_stream(side)
    assert float(out64.abs().sum()) == 0.0
    g.replay()
    torch.cuda.synchronize()
    chk = launches - 1
    ref = oracle.batch_spectra_u8(iq_host[chk], N, nthreads=8)
    assert rel_err(out64[chk].cpu().numpy(), ref, EPS_STRICT).max() <= 1e-10
    assert torch.equal(out32, out64.to(torch.float32))
    eng.close()
