"""bench.py's host-side blocks on CPU tensors (the GPU step itself needs a device):
the parity block's metrics and the cpu_baseline block's bookkeeping, so that a slip in
either shows here and not only on the GPU box."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rtl-ws_amd"))


def test_workload_table_is_consistent():
    import bench
    assert bench.HEADLINE in bench.WORKLOADS and set(bench.EXTRA_WORKLOADS) <= set(bench.WORKLOADS)
    for name, (n_fft, k_avg, window, output, cic_r, frames) in bench.WORKLOADS.items():
        assert frames % k_avg == 0 and n_fft in (1024, 2048, 4096), name
        assert window in ("rect", "hann") and output in ("power_sum", "mean_db", "payload_u8", "cs32"), name
    # SURVEY.md §8d's contract figures
    assert bench.algorithmic_bytes_per_frame(1024, 1, 0) * 65536 == 402653184
    assert bench.algorithmic_bytes_per_frame(4096, 8, 0) == 10240
    assert bench.algorithmic_bytes_per_frame(2048, 1, 8) == 40960
    assert bench.algorithmic_bytes_per_frame(2048, 1, 12) == 2 * 2048 * 12 + 4 * 2048
    assert bench.algorithmic_bytes_per_frame(1024, 6, 0, "payload_u8") == 2048 + 1024 // 6
    assert bench.algorithmic_bytes_per_frame(2048, 1, 8, "cs32") == 2 * 2048 * 8 + 8 * 2048


@pytest.mark.parametrize("name", ["batched_1024pt_64k_frames", "hann_4096pt_k8_db", "cic8_2048pt",
                                  "k6_1024pt_payload", "cic8_block_sums"])
def test_parity_block_on_oracle_output(oracle, name):
    """Feed the parity block the oracle's own result (rounded to what the kernel stores):
    every metric it reports must then be at rounding level."""
    import bench
    from rtlws import synth
    wl = bench.WORKLOADS[name]
    n_fft, k_avg, window, output, cic_r, _ = wl
    rows = 4
    nchk = rows * k_avg
    iq = synth.tone_noise_iq(nchk, n_fft * max(cic_r, 1), seed=3)
    if output == "cs32":
        got = (iq.astype(np.int32) - 128).reshape(-1, cic_r, 2).sum(axis=1).reshape(nchk, -1)
        assert bench.parity_block(np, oracle, wl, iq, got, nchk) == {"frames": nchk, "bit_exact": True}
        return
    w = None if window == "rect" else (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft))
    ref = (oracle.batch_spectra_cic_u8(iq, n_fft, cic_r, K=k_avg) if cic_r > 1
           else oracle.batch_spectra_u8(iq, n_fft, K=k_avg, window=w))
    if output == "mean_db":
        got = (10 * np.log10(ref / k_avg)).astype(np.float32)
        out = bench.parity_block(np, oracle, wl, iq, got, nchk)
        assert out["frames"] == nchk and out["max_abs_db_err"] < 1e-5
    elif output == "payload_u8":
        got = np.stack([oracle.spectrum_payload(r, k_avg, 0) for r in ref])
        out = bench.parity_block(np, oracle, wl, iq, got, nchk)
        assert out["bytes_differing"] == 0 and out["bytes"] == rows * n_fft
    else:
        out = bench.parity_block(np, oracle, wl, iq, ref.astype(np.float32), nchk)
        assert out["max_rel_err_floor1e-9"] < 1e-6 and out["p99.9_rel_err_floor1e-9"] < 1e-6
        assert out["max_rel_err_floor1e-5"] <= out["max_rel_err_floor1e-9"]


def test_cpu_baseline_block_bookkeeping(oracle):
    import torch
    import bench
    from rtlws import synth
    wl = bench.WORKLOADS["batched_1024pt_64k_frames"]
    dev_in = torch.from_numpy(synth.tone_noise_iq(256, 1024, seed=5))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 256, budget_scale=0.02)
    assert out["unit"] == "spectra/s" and out["kind"] == "port" and out["nproc"] == os.cpu_count()
    assert 1 <= out["cores"] <= 16 and out["value"] > 0
    assert out["one_thread"]["cores"] == 1 and out["one_thread"]["value"] > 0
    wl = bench.WORKLOADS["cic8_block_sums"]
    dev_in = torch.from_numpy(synth.uniform_iq(8, 2048 * 8, seed=6))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 8, budget_scale=0.02)
    assert out["unit"] == "samples/s" and out["cores"] == 1 and out["value"] > 0
