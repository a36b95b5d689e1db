"""rtlws_multi.h on the GPU box (one device): a batch through the sharded C path must be the SAME
rows, bit for bit, as one rtlws_spectra_batch launch over the whole batch -- with one shard, and
with several shards that all live on device 0 (the N-shard code path: partition, per-shard engines
and threads, concatenation; not a scaling measurement) -- in every arithmetic, with K-groups that do
not divide evenly and with a ragged tail."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["f32", "f64", "f64c_f32o"])
@pytest.mark.parametrize("shards", [1, 3])
@pytest.mark.parametrize("K,window,output", [(1, "rect", "power_sum"), (6, "hann", "mean_db"), (8, "rect", "payload_u8")])
def test_sharded_batch_equals_one_launch(engine, built, prec, shards, K, window, output):
    from rtlws import synth
    N = 1024
    B = 1000 * K + (K - 1)                         # a ragged tail of K-1 frames that no shard owns
    iq = synth.tone_noise_iq(B, N, seed=5 + K)
    f64 = prec != "f32"
    flags = built.FLAG_ROWS_F32 if prec == "f64c_f32o" else 0
    desc = built.make_desc(N, K, "cu8", window, output, 0, 15, flags)
    mb = built.MultiBatch(desc, B, device_ids=[0] * shards, f64=f64)
    assert mb.shards == shards and mb.frames == 1000 * K
    mb.upload(iq)
    stats, wall = mb.run(3)
    got = mb.download()
    mb.close()
    want = engine.spectra(iq[:1000 * K], N, k_avg=K, window=window, output=output, gain_db=15, f64=f64,
                          rows_f32=(prec == "f64c_f32o"))
    assert got.dtype == want.dtype and got.shape == want.shape == (1000, N)
    assert np.array_equal(got, want, equal_nan=True)
    assert [s.device for s in stats] == [0] * shards and all(s.rc == 0 and s.launches == 3 for s in stats)
    assert sum(s.frames for s in stats) == 1000 * K and all(s.frames % K == 0 for s in stats)
    assert all(s.event_ms > 0 and s.wall_ms >= s.event_ms * 0.5 for s in stats) and wall >= max(s.wall_ms for s in stats) - 1e-9


@pytest.mark.parametrize("f64", [False, True])
def test_sharded_cic_fused_batch(engine, built, f64):
    """configs[3]'s shape through the sharded path: raw IQ, CIC 8:1 + 2048-point, 3 shards on device 0."""
    from rtlws import synth
    N, R, B = 2048, 8, 50
    iq = synth.tone_noise_iq(B * R, N, seed=9).reshape(B, N * R, 2)
    desc = built.make_desc(N, 1, "cu8", "rect", "power_sum", R)
    mb = built.MultiBatch(desc, B, device_ids=[0, 0, 0], f64=f64)
    mb.upload(iq)
    mb.run(2)
    got = mb.download()
    mb.close()
    want = engine.spectra(iq, N, cic_r=R, f64=f64)
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_more_shards_than_rows_and_empty_batch(engine, built):
    from rtlws import synth
    iq = synth.tone_noise_iq(2, 1024, seed=1)
    mb = built.MultiBatch(built.make_desc(1024), 2, device_ids=[0, 0, 0, 0])
    mb.upload(iq)
    stats, _ = mb.run(1)
    got = mb.download()
    mb.close()
    assert sorted(s.frames for s in stats) == [0, 0, 1, 1]
    assert np.array_equal(got, engine.spectra(iq, 1024))
    mb = built.MultiBatch(built.make_desc(1024), 0, device_ids=[0])
    mb.upload(np.zeros(0, dtype=np.uint8))
    mb.run(2)
    assert mb.download().shape == (0, 1024)
    mb.close()


def test_driver_line_on_this_box(built):
    """rtlws_multi_batch on whatever devices the box has, and the 2-shard rehearsal on device 0."""
    exe = os.path.join(built.LIB_DIR, "rtlws_multi_batch")
    for extra, shards in (([], built.device_count()), (["--shards-on-device0", "2"], 2),
                          (["--shards-per-device", "2"], 2 * built.device_count())):
        out = subprocess.run([exe, "--frames", "8192", "--launches", "20", "--warmup", "20"] + extra,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        r = json.loads(out.stdout)
        assert r["shards"] == shards and r["frames_used"] == 8192 and r["spectra_per_s_total"] > 1e6
        assert r["rehearsal_all_on_device0"] == ("--shards-on-device0" in extra)
        if "--shards-per-device" in extra:
            assert [s["device"] for s in r["per_shard"]] == [g // 2 for g in range(shards)]
        assert sum(s["frames"] for s in r["per_shard"]) == 8192
        assert all(s["event_ms_per_launch"] > 0 and s["spectra_per_s"] > 0 for s in r["per_shard"])


def test_bench_wrapper_line(built):
    """bench.py --workload multi_batch: the C driver's result in the bench line's shape."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "multi_batch", "--steps", "50",
                          "--frames", "8192"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["config"]["workload"] == "multi_batch" and r["n_gpus"] == built.device_count() and r["scaling"] == "weak"
    assert r["config"]["frames_per_gpu"] == 8192 and r["config"]["frames_per_step"] == 8192 * r["n_gpus"]
    assert 0 < r["roofline"]["frac"] < 1 and len(r["per_device"]) == r["n_gpus"] and r["value"] > 1e6
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "multi_batch", "--steps", "50",
                          "--frames", "8192", "--shards-per-device", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r2 = json.loads(out.stdout.strip().splitlines()[-1])
    assert r2["n_gpus"] == built.device_count() and len(r2["per_device"]) == 2 * r2["n_gpus"]
    assert r2["config"]["shards_per_device"] == 2 and "not one kernel's duration" in r2["roofline"]["frac_clock"]
