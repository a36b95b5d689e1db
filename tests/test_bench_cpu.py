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
    assert bench.algorithmic_bytes_per_frame(1024, 1, 0, "power_sum", f64=True) == 10240   # 2N + 8N/K
    assert set(bench.F64_WORKLOADS) <= set(bench.WORKLOADS) and set(bench.F64C_F32O_WORKLOADS) <= set(bench.WORKLOADS)
    assert not set(bench.F64_WORKLOADS) & set(bench.F64C_F32O_WORKLOADS)
    assert bench.algorithmic_bytes_per_frame(2048, 1, 8, "power_sum", f64=True) == 49152           # configs[3], f64 rows
    assert bench.precision_of("batched_1024pt_64k_frames_f64c_f32o") == "f64c_f32o"
    assert bench.precision_of("cic8_2048pt_f64") == "f64" and bench.precision_of("cic8_2048pt") == "f32"
    assert bench.parity_bounds_for("cic8_2048pt") is None
    assert bench.parity_bounds_for("cic8_2048pt_f64")["max_rel_err_floor1e-9"] == 1e-10
    assert bench.parity_bounds_for("cic8_2048pt_f64c_f32o")["max_rel_err_floor1e-9"] <= 6e-8
    # every BASELINE.json configuration on the default line has the CPU path timed beside it
    # ... in f32 AND in the reference's arithmetic: configs[2] and configs[3]
    assert set(bench.EXTRA_CPU_BASELINE) == {"hann_4096pt_k8_db", "hann_4096pt_k8_db_f64c_f32o", "cic8_2048pt",
                                             "cic8_2048pt_f64"} <= set(bench.EXTRA_WORKLOADS)
    assert bench.precision_of("hann_4096pt_k8_db_f64c_f32o") == "f64c_f32o"
    assert bench.algorithmic_bytes_per_frame(4096, 8, 0, "mean_db", f64=False) == 10240      # configs[2], f32 rows


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
    mask = len(os.sched_getaffinity(0))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 256, budget_scale=0.02, quota=None)
    assert out["unit"] == "spectra/s" and out["kind"] == "port" and out["nproc"] == os.cpu_count()
    assert out["affinity_mask_cores"] == mask and out["cgroup_cpu_quota"] is None
    assert [r["cores"] for r in out["runs"]] == sorted({1, mask})
    # `value` is the fastest run and `cores` the thread count that produced it
    best = max(out["runs"], key=lambda r: r["value"])
    assert out["value"] == best["value"] and out["cores"] == best["cores"] and out["value"] > 0
    assert out["one_thread"]["cores"] == 1 and out["one_thread"]["value"] > 0
    # a faked quota smaller than the mask: timed with 1, the quota and the mask
    if mask >= 3:
        out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 256, budget_scale=0.02, quota=2.0)
        assert [r["cores"] for r in out["runs"]] == [1, 2, mask] and out["cgroup_cpu_quota"] == 2.0
        best = max(out["runs"], key=lambda r: r["value"])
        assert (out["value"], out["cores"]) == (best["value"], best["cores"])
    wl = bench.WORKLOADS["cic8_block_sums"]
    dev_in = torch.from_numpy(synth.uniform_iq(8, 2048 * 8, seed=6))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 8, budget_scale=0.02)
    assert out["unit"] == "samples/s" and out["cores"] == 1 and out["value"] > 0


def test_cpu_quota_is_read_from_the_cgroup(tmp_path):
    """the job's CPU share is read (cpu.max of cgroup v2, the cfs pair of v1, the
    smallest over the process's cgroup and its ancestors), not inferred."""
    import bench
    root = tmp_path / "cg"
    (root / "jobs" / "j1").mkdir(parents=True)
    proc = tmp_path / "cgroup"
    proc.write_text("0::/jobs/j1\n")
    assert bench.cpu_quota(str(root), str(proc)) is None                       # nothing to read
    (root / "cpu.max").write_text("max 100000\n")
    assert bench.cpu_quota(str(root), str(proc)) is None                       # unlimited
    (root / "jobs" / "j1" / "cpu.max").write_text("1600000 100000\n")
    assert bench.cpu_quota(str(root), str(proc)) == 16.0
    (root / "jobs" / "cpu.max").write_text("800000 100000\n")                  # a tighter ancestor wins
    assert bench.cpu_quota(str(root), str(proc)) == 8.0
    # cgroup v1: cpu controller mounted under <root>/cpu
    root1 = tmp_path / "cg1"
    (root1 / "cpu" / "pod").mkdir(parents=True)
    proc1 = tmp_path / "cgroup1"
    proc1.write_text("4:memory:/x\n1:cpu,cpuacct:/pod\n")
    (root1 / "cpu" / "pod" / "cpu.cfs_quota_us").write_text("-1\n")
    (root1 / "cpu" / "pod" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cpu_quota(str(root1), str(proc1)) is None
    (root1 / "cpu" / "pod" / "cpu.cfs_quota_us").write_text("250000\n")
    assert bench.cpu_quota(str(root1), str(proc1)) == 2.5
    assert bench.cpu_thread_counts(256, 16.0) == [1, 16, 256]
    assert bench.cpu_thread_counts(256, 2.5) == [1, 3, 256]
    assert bench.cpu_thread_counts(8, None) == [1, 8] and bench.cpu_thread_counts(8, 64.0) == [1, 8]
    assert bench.cpu_thread_counts(1, 0.5) == [1]


def test_headline_is_the_reference_arithmetic_workload():
    """the driver parses the default line -- it must be configs[1] in the reference's
    arithmetic (f64, src/spectrum.c:21,28,54-58) at the contract's 6 144 B per spectrum; the f32 kernel on the
    same frames rides along as the fast mode."""
    import bench
    assert bench.HEADLINE == "batched_1024pt_64k_frames_f64c_f32o" and bench.precision_of(bench.HEADLINE) == "f64c_f32o"
    assert bench.WORKLOADS[bench.HEADLINE] == (1024, 1, "rect", "power_sum", 0, 65536)
    assert bench.algorithmic_bytes_per_frame(1024, 1, 0, "power_sum", f64=False) == 6144      # f32 rows
    assert bench.FAST_MODE in bench.EXTRA_WORKLOADS and bench.precision_of(bench.FAST_MODE) == "f32"
    assert bench.HEADLINE not in bench.EXTRA_WORKLOADS
    assert not hasattr(bench, "EXTRA_SPLIT")         # the "split" option was a measured loss: gone in round 6
    assert bench.parity_bounds_for(bench.HEADLINE)["max_rel_err_floor1e-9"] <= 6e-8


def test_cpu_baseline_block_windowed_and_cic_cases(oracle, monkeypatch):
    """configs[2] (Hann, K = 8, mean dB) and configs[3] (CIC 8:1 + 2048-point): the window and the
    decimator reach the oracle call that is timed, K-groups stay whole, the sample text says so."""
    import torch
    import bench
    from rtlws import synth
    seen = []
    real_u8, real_cic = oracle.batch_spectra_u8, oracle.batch_spectra_cic_u8
    monkeypatch.setattr(oracle, "batch_spectra_u8", lambda h, n, **kw: (seen.append(("u8", n, kw)), real_u8(h, n, **kw))[1])
    monkeypatch.setattr(oracle, "batch_spectra_cic_u8", lambda h, n, r, **kw: (seen.append(("cic", n, r, kw)), real_cic(h, n, r, **kw))[1])
    wl = bench.WORKLOADS["hann_4096pt_k8_db"]
    dev_in = torch.from_numpy(synth.tone_noise_iq(64, 4096, seed=5))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 64, budget_scale=0.01, quota=None)
    assert out["value"] > 0 and out["one_thread"]["value"] > 0 and "Hann, mean dB" in out["sample"] and "K = 8" in out["sample"]
    assert seen and all(k[0] == "u8" and k[1] == 4096 and k[2]["K"] == 8 for k in seen)
    assert all(np.allclose(k[2]["window"], synth.hann(4096)) for k in seen)
    assert all(k[2]["out"].shape[0] * 8 <= 64 for k in seen)
    seen.clear()
    wl = bench.WORKLOADS["cic8_2048pt"]
    dev_in = torch.from_numpy(synth.uniform_iq(16, 2048 * 8, seed=6))
    out = bench.cpu_baseline_block(np, oracle, wl, dev_in, 16, budget_scale=0.01, quota=None)
    assert out["value"] > 0 and "CIC 8:1 first" in out["sample"]
    assert seen and all(k[0] == "cic" and k[1] == 2048 and k[2] == 8 and k[3]["window"] is None for k in seen)


def test_parity_block_names_constant_input_rows_instead_of_dividing_by_zero(oracle):
    """the driver's cic12 block was 0/0 = NaN with exit code 0.  A constant
    input frame (all-zero oracle row) must be reported as such and fail the run."""
    import bench
    from rtlws import synth
    wl = bench.WORKLOADS["cic12_2048pt"]
    iq = synth.tone_noise_iq(4, 2048 * 12, seed=9)
    iq[2] = 255                                    # what a clobbered synthesis temporary produces
    ref = oracle.batch_spectra_cic_u8(iq, 2048, 12)
    with np.errstate(all="raise"):                 # no 0/0 anywhere
        out = bench.parity_block(np, oracle, wl, iq, ref.astype(np.float32), 4)
    assert out["non_finite"] and out["constant_input_rows"] == [2]
    assert out["input_byte_min_max_of_those_rows"] == [[255, 255]]
    assert bench.parity_failures(out) == ["constant_input_rows"]


def test_parity_failures_and_strict_json():
    import json
    import bench
    ok = {"frames": 256, "max_rel_err_floor1e-5": 4e-5, "max_rel_err_floor1e-9": 4e-4, "p99.9_rel_err_floor1e-9": 3e-5}
    assert bench.parity_failures(ok) == []
    assert bench.parity_failures(dict(ok, **{"p99.9_rel_err_floor1e-9": 2e-4})) == ["p99.9_rel_err_floor1e-9"]
    assert bench.parity_failures(dict(ok, **{"max_rel_err_floor1e-9": float("nan")})) == ["max_rel_err_floor1e-9"]
    assert bench.parity_failures({"frames": 8, "bit_exact": False}) == ["bit_exact"]
    assert bench.parity_failures({"frames": 8, "max_abs_db_err": 1e-3}) == ["max_abs_db_err"]
    assert bench.parity_failures({"max_rel_err_floor1e-9": 3e-5}, bench.PARITY_BOUNDS_F64) == ["max_rel_err_floor1e-9"]
    line = json.dumps(bench.strict_json({"a": float("nan"), "b": [1.0, float("inf")], "c": {"d": 2}}), allow_nan=False)
    assert json.loads(line) == {"a": "nan", "b": [1.0, "inf"], "c": {"d": 2}}


def test_valu_issue_frac_arithmetic(tmp_path, monkeypatch):
    import json
    import bench
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "valu_insts.json").write_text(json.dumps(
        {"w": {"issue_cycles_per_launch": 8.0e7, "sclk_ghz_under_load": 2.0, "frames_per_launch": 65536}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out = bench.valu_issue_frac("w", 80e-6, 256)
    # 8e7 issue cycles over 4 SIMDs x 256 CUs x 80 us x 2 GHz
    assert abs(out["valu_issue_frac"] - 8.0e7 / (1024 * 80e-6 * 2.0e9)) < 1e-12
    assert "ANOTHER launch series" in out["valu_issue_source"]
    own = bench.valu_issue_frac("w", 80e-6, 256, sclk_ghz=2.2)      # the run's own clock: what a line may carry
    assert abs(own["valu_issue_frac"] - 8.0e7 / (1024 * 80e-6 * 2.2e9)) < 1e-12 and "own shader clock" in own["valu_issue_source"]
    half = bench.valu_issue_frac("w", 40e-6, 256, sclk_ghz=2.2, frames=32768)     # half the frames in half the time
    assert abs(half["valu_issue_frac"] - own["valu_issue_frac"]) < 1e-12
    assert bench.valu_issue_frac("other", 80e-6, 256) is None


def test_parity_block_single_frame_db_rows_are_bounded_within_50_db(oracle):
    """K = 1 dB rows of the f32 kernel: the bound applies to bins within 50 dB of the row maximum
    (DESIGN.md §5); an error placed on a weak bin is reported under `all_bins` and does not fail,
    the same error on a strong bin does."""
    import bench
    from rtlws import synth
    wl = (1024, 1, "rect", "mean_db", 0, 4)
    iq = synth.pure_tone_iq(4, 1024, seed=1)
    ref = oracle.batch_spectra_u8(iq, 1024)
    db = (10 * np.log10(ref)).astype(np.float32)
    weak = int(np.argmin(ref[0]))
    strong = int(np.argmax(ref[0]))
    assert ref[0, weak] < 1e-5 * ref[0, strong]
    bad_weak = db.copy()
    bad_weak[0, weak] += 0.01
    out = bench.parity_block(np, oracle, wl, iq, bad_weak, 4)
    assert out["max_abs_db_err_all_bins"] > 5e-3 and out["max_abs_db_err_within_50db"] < 1e-4
    assert bench.parity_failures(out) == []
    bad_strong = db.copy()
    bad_strong[0, strong] += 0.01
    out = bench.parity_block(np, oracle, wl, iq, bad_strong, 4)
    assert bench.parity_failures(out) == ["max_abs_db_err_within_50db"]
    # f64 rows are bounded on every bin
    out = bench.parity_block(np, oracle, wl, iq, 10 * np.log10(ref), 4, f64=True)
    assert out["max_abs_db_err"] < 1e-12


def test_rank_pinning_helper(built):
    """bench.py under N > 1: every rank keeps to its own GPU's NUMA node (include/rtlws_topo.h); here with a faked
    description, in a child process so that this one keeps its mask."""
    import subprocess
    code = """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import bench, rtlws
assert bench.cpus_of_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and bench.cpus_of_cpulist("") == []
allowed = sorted(os.sched_getaffinity(0))
keep = allowed[:max(1, len(allowed) // 2)]
class T: numa_node = 1; bus_id = b"0000:15:00.0"; cpulist = (",".join(map(str, keep)) + ",4000").encode()
info = bench.pin_rank_to_its_gpu(rtlws, 1, describe=lambda d: T)
assert info == {"numa_node": 1, "bus_id": "0000:15:00.0", "cpus_pinned": len(keep)}, info
assert sorted(os.sched_getaffinity(0)) == keep
class Far: numa_node = 0; bus_id = b"x"; cpulist = b"4001-4005"
assert bench.pin_rank_to_its_gpu(rtlws, 0, describe=lambda d: Far)["cpus_pinned"] == 0
assert sorted(os.sched_getaffinity(0)) == keep
assert bench.pin_rank_to_its_gpu(rtlws, 0, describe=lambda d: None) == {"numa_node": -1, "bus_id": "", "cpus_pinned": 0}
print("ok")
""" % (ROOT, os.path.join(ROOT, "rtl-ws_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_energy_counter_is_found_by_pci_bus_id():
    """rocm_smi numbers every physical GPU and ignores HIP_VISIBLE_DEVICES, so the package
    energy accumulator is looked up by the bus id HIP reports -- against a fake rsmi here: the third device matches,
    a partition id in bits 28-31 does not matter, an unknown bus id gives None (the energy leg is dropped)."""
    import ctypes as C
    from rtlws import energy

    class FakeSmi:
        ids = [(0 << 32) | (0x05 << 8), (0 << 32) | (0x15 << 8), (3 << 28) | (0 << 32) | (0x65 << 8) | (0 << 3) | 0]

        def rsmi_num_monitor_devices(self, n):
            n._obj.value = len(self.ids)
            return 0

        def rsmi_dev_pci_id_get(self, i, out):
            out._obj.value = self.ids[i.value]
            return 0

        def rsmi_dev_energy_count_get(self, i, cnt, res, ts):
            cnt._obj.value, res._obj.value = 1000 * (i.value + 1), 15.3
            return 0

    assert energy.parse_bus_id("0000:0d:00.0") == (0, 0x0d, 0, 0) and energy.parse_bus_id("nonsense") is None
    assert energy.bdf_fields((1 << 32) | (0x65 << 8) | (2 << 3) | 1) == (1, 0x65, 2, 1)
    assert energy.rsmi_index_for_bus_id("0000:65:00.0", FakeSmi()) == 2
    assert energy.rsmi_index_for_bus_id("0000:05:00.0", FakeSmi()) == 0
    assert energy.rsmi_index_for_bus_id("0000:99:00.0", FakeSmi()) is None
    ec = energy.EnergyCounter("0000:65:00.0", FakeSmi())
    assert ec.index == 2 and abs(ec.joules() - 3000 * 15.3e-6) < 1e-9
    assert energy.EnergyCounter("0000:99:00.0", FakeSmi()).joules() is None
