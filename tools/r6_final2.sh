#!/bin/bash
# after the Hann weights moved into the first butterfly layer: the GPU suite, the default line in both forms, the
# rocprofv3 passes of the workload whose kernel changed
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; rc=$?
tail -5 gpurun_out/r06_gpu_suite.txt
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/r06_gpu_suite.txt | head -30; exit $rc; }
timeout -k 10 60 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python3 bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default_line.err || { tail -20 gpurun_out/r06_bench_default_line.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default_line_steps20.json 2> gpurun_out/r06_bench_steps20.err || { tail -20 gpurun_out/r06_bench_steps20.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/r06_bench_default_line.json", "gpurun_out/r06_bench_default_line_steps20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f)
    def show(name, x, r):
        en = r.get("energy") or {}
        print("  %-38s frac %.4f wall %.4f %8.2f us  mJ %s W %s sclk %s" % (name, r["frac"], r["frac_wall"], r["avg_launch_us"],
              ("%.1f" % en["mj_per_launch"]) if en else None, ("%.0f" % en["watts"]) if en else None, ("%.3f" % r["sclk_ghz"]) if r.get("sclk_ghz") else None))
    show(d["config"]["workload"], d, r)
    print("  box", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get("box") or {}).items() if k != "kernel"})
    for x in d.get("extra_workloads", []):
        show(x["workload"], x, x["roofline"])
PY
for wl in hann_4096pt_k8_db_f64c_f32o; do
  bash tools/profile_gpu.sh r06_$wl $wl > gpurun_out/prof_r06_$wl.log 2>&1
  echo "profiled $wl: $(ls gpurun_out/prof_r06_$wl | grep -c summary) summaries"
done
