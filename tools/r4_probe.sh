#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_bench_gpu.py -x -q > gpurun_out/r04_probe.pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04_probe.pytest.log
python3 bench.py > gpurun_out/r04_bench_default_line_probe.json 2> gpurun_out/r04_bench_probe.err; echo "rc=$?"
python3 bench.py --steps 20 > gpurun_out/r04_bench_default_line_probe_steps20.json 2>> gpurun_out/r04_bench_probe.err; echo "rc=$?"
python3 bench.py --no-clock-probe --no-extra --no-cpu-baseline > gpurun_out/r04_bench_noprobe.json 2>> gpurun_out/r04_bench_probe.err
python3 bench.py --no-clock-probe --no-extra --no-cpu-baseline --steps 20 > gpurun_out/r04_bench_noprobe_steps20.json 2>> gpurun_out/r04_bench_probe.err
python3 - <<'PY'
import json
for f in ("r04_bench_default_line_probe","r04_bench_default_line_probe_steps20","r04_bench_noprobe","r04_bench_noprobe_steps20"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "value %.4g frac %.4f wall %.4f sclk %s valu %s" % (d["value"], r["frac"], r["frac_wall"], r.get("sclk_ghz"), r.get("valu_issue_frac")))
    for x in d.get("extra_workloads", []):
        print("   ", x["workload"], "%.4f" % x["roofline"]["frac"], x["roofline"].get("sclk_ghz"), x["roofline"].get("valu_issue_frac"))
PY
