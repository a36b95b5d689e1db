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
    for name in ("batched_1024pt_64k_frames", "hann_4096pt_k8_db", "cic12_2048pt", "cic8_2048pt",
                 "batched_1024pt_64k_frames_f64"):
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
