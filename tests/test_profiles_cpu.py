"""The committed rocprof evidence agrees with the bench line it was taken with
(VERDICT r1 "close the evidence chain"): for every profiles/r02_*_timed_launches.json,
algorithmic bytes / (rocprofv3 per-dispatch average over the TIMED launches) is within
3 % of the line's roofline.achieved, and that average does not exceed ms_per_step."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_average_matches_bench_events():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_*_timed_launches.json")))
    assert files, "no timed-launch summaries committed"
    seen = set()
    for f in files:
        d = json.load(open(f))
        b = d["bench_line_same_run"]
        assert d["dropped_first"] >= 500 and d["timed_launches"] == b["steps"]
        achieved_from_trace = b["roofline"]["algorithmic_bytes_per_launch"] / d["timed_avg_ns"]
        assert abs(achieved_from_trace / b["roofline"]["achieved"] - 1.0) <= 0.03, f
        assert d["timed_avg_ns"] <= 1e6 * b["ms_per_step"], f
        seen.add(d["workload"])
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.HEADLINE in seen and bench.FAST_MODE in seen      # the headline (and the fast mode) must be among them


def test_hbm_traffic_is_close_to_algorithmic():
    """No wasted re-reads / write amplification on the measured workloads (<= 3 % over)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    # (round 6: every BASELINE configuration in the reference's arithmetic is among them -- configs[1] headline,
    # configs[2] hann_4096pt_k8_db_f64c_f32o, configs[3] cic8_2048pt_f64 -- from this round's own --pmc passes)
    for name in ("batched_1024pt_64k_frames", "hann_4096pt_k8_db", "cic12_2048pt", "cic8_2048pt",
                 "batched_1024pt_64k_frames_f64", "batched_1024pt_64k_frames_f64c_f32o",
                 "hann_4096pt_k8_db_f64c_f32o", "cic8_2048pt_f64"):
        n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
        alg = bench.algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output, name in bench.F64_WORKLOADS) * frames
        assert 0.97 <= t[name]["bytes_per_launch"] / alg <= 1.03, (name, t[name]["bytes_per_launch"], alg)


def test_valu_issue_fractions_are_plausible():
    """profiles/valu_insts.json x the committed launch durations: between a third and two thirds of
    the SIMDs' issue cycles on the f32 FFT kernels (none of them is issue-bound)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for name, lo, hi in (("hann_4096pt_k8_db", 0.45, 0.65), ("batched_1024pt_64k_frames", 0.38, 0.60),
                         ("batched_1024pt_64k_frames_f64", 0.30, 0.60)):
        t = json.load(open(os.path.join(ROOT, "profiles", "r03_%s_timed_launches.json" % name)))
        v = bench.valu_issue_frac(name, t["timed_avg_ns"] * 1e-9, 256)
        assert v is not None and lo < v["valu_issue_frac"] < hi, (name, v)


def test_round6_profiles_cover_every_baseline_config_in_the_reference_arithmetic():
    """the fraction of every BASELINE config can be recomputed from THIS round's profiles --
    per-dispatch average over the timed launches (rocprofv3 --kernel-trace) and the --pmc traffic."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for name, lo, hi in (("batched_1024pt_64k_frames_f64c_f32o", 0.43, 0.52), ("hann_4096pt_k8_db_f64c_f32o", 0.12, 0.16),
                         ("hann_4096pt_k8_db", 0.25, 0.31), ("cic8_2048pt_f64", 0.65, 0.74)):
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_%s_timed_launches.json" % name)))
        p = json.load(open(os.path.join(ROOT, "profiles", "r06_%s_pmc.json" % name)))
        n_fft, k_avg, window, output, cic_r, frames = bench.WORKLOADS[name]
        alg = bench.algorithmic_bytes_per_frame(n_fft, k_avg, cic_r, output, name in bench.F64_WORKLOADS) * frames
        frac = alg / d["timed_avg_ns"] / bench.HBM_PEAK_GBS
        assert lo < frac < hi, (name, frac)
        assert d["timed_launches"] == 2000 and d["dropped_first"] == 500
        assert p["hbm"]["bytes_per_launch"] == t[name]["bytes_per_launch"]
        # the bench line of the profiled run carries no probe wavefront beside its timed launches
        assert "sclk_ghz" not in d["bench_line_same_run"]["roofline"]
