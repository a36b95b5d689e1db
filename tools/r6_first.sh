#!/bin/bash
# round 6, first GPU call: the whole GPU suite on the restructured build (hidden visibility, no split, no hooks in the
# product sources), the default line in both forms, and the energy of the three configs whose r06 profiles follow
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; rc=$?
tail -5 gpurun_out/r06_gpu_suite.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default_line.err || { tail -20 gpurun_out/r06_bench_default_line.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default_line_steps20.json 2> gpurun_out/r06_bench_steps20.err || { tail -20 gpurun_out/r06_bench_steps20.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/r06_bench_default_line.json", "gpurun_out/r06_bench_default_line_steps20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f)
    print("  %-40s frac %.4f wall %.4f  %.2f us  mJ %s  W %s  sclk %s  cpu %.3g/%d" % (d["config"]["workload"], r["frac"], r["frac_wall"], r["avg_launch_us"],
          r.get("energy", {}).get("mj_per_launch"), r.get("energy", {}).get("watts"), r.get("sclk_ghz"), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
    print("  box", r.get("box"))
    for x in d.get("extra_workloads", []):
        r = x["roofline"]
        print("  %-40s frac %.4f wall %.4f  %.2f us  mJ %s  W %s  sclk %s  cpu %s  parity %s" % (x["workload"], r["frac"], r["frac_wall"], r["avg_launch_us"],
              r.get("energy", {}).get("mj_per_launch"), r.get("energy", {}).get("watts"), r.get("sclk_ghz"),
              ("%.3g/%d" % (x["cpu_baseline"]["value"], x["cpu_baseline"]["cores"])) if "cpu_baseline" in x else "-", x["parity"]))
PY
OUT=gpurun_out/r06_energy_configs.txt; : > $OUT
for rep in 1 2; do
for wl in hann_4096pt_k8_db_f64c_f32o hann_4096pt_k8_db_f64 hann_4096pt_k8_db rect_4096pt_f64 cic8_2048pt_f64 batched_1024pt_64k_frames_f64c_f32o; do
timeout -k 10 120 python3 tools/energy_per_launch.py $wl 6000 2>/dev/null >> $OUT || echo "$wl FAILED" >> $OUT
done; done
cat $OUT
