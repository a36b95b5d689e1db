#!/bin/bash
# round-6 closing soak (one gpurun call): the whole GPU suite again, longer fuzz seeds, the > 4 GiB batch, smoke(),
# the C drivers (multi_batch, eight paced sensors), the drop-in latency, the 1 Mi-frame and uniform-input side runs
OUT=gpurun_out/r06_soak.txt; : > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_soak_gputests.log 2>&1; echo "pytest -m gpu rc $?: $(tail -1 gpurun_out/r06_soak_gputests.log)" >> $OUT
for s in 621 622 623; do timeout -k 10 200 python3 tests/tools/fuzz_parity.py $s 60 2>&1 | tail -1 >> $OUT; done
for s in 71 72 73 74; do timeout -k 10 200 python3 tests/tools/fuzz_parity_f64.py $s 60 2>&1 | tail -1 >> $OUT; done
timeout -k 10 300 python3 tests/tools/big_batch_check.py 2>&1 | tail -2 >> $OUT
python3 -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
timeout -k 10 200 python3 bench.py --workload multi_batch --steps 500 > gpurun_out/r06_multi_batch.jsonl 2> gpurun_out/r06_multi_batch.err || echo "multi_batch FAILED" >> $OUT
timeout -k 10 200 python3 bench.py --workload multi_batch --steps 500 --shards-per-device 2 >> gpurun_out/r06_multi_batch.jsonl 2>> gpurun_out/r06_multi_batch.err || echo "multi_batch 2 FAILED" >> $OUT
timeout -k 10 200 python3 bench.py --workload realtime_8x2400k --steps 4000 > gpurun_out/r06_multi_stream.jsonl 2> gpurun_out/r06_multi_stream.err || echo "realtime FAILED" >> $OUT
timeout -k 10 200 python3 tools/dropin_latency.py > gpurun_out/r06_dropin_latency.txt 2>&1 || echo "dropin latency FAILED" >> $OUT
timeout -k 10 300 python3 bench.py --frames 1048576 --sets 2 --steps 200 --no-cpu-baseline --no-extra > gpurun_out/r06_bench_1Mi_frames.json 2> gpurun_out/r06_bench_1Mi.err || echo "1Mi FAILED" >> $OUT
timeout -k 10 300 python3 bench.py --input uniform --no-cpu-baseline --no-extra > gpurun_out/r06_bench_uniform_input.json 2> gpurun_out/r06_bench_uniform.err || echo "uniform FAILED" >> $OUT
python3 - >> $OUT <<'PY'
import json
for f in ("gpurun_out/r06_bench_1Mi_frames.json", "gpurun_out/r06_bench_uniform_input.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; e = r.get("energy") or {}
        print("%s: %.3g spectra/s, frac %.4f (%.1f us), %.1f mJ per launch, parity %s" % (f, d["value"], r["frac"], r["avg_launch_us"], e.get("mj_per_launch", 0), d["parity"]))
    except Exception as ex:
        print(f, "unreadable:", ex)
for f in ("gpurun_out/r06_multi_batch.jsonl", "gpurun_out/r06_multi_stream.jsonl"):
    for ln in open(f).read().strip().splitlines():
        d = json.loads(ln)
        print("%s: %s value %.4g n_gpus %s frac %s realtime %s" % (f, d["config"]["workload"], d["value"], d["n_gpus"], (d.get("roofline") or {}).get("frac"),
              {k: d["realtime"][k] for k in ("chunks_dropped", "latency_ms_max")} if "realtime" in d else None))
PY
cat gpurun_out/r06_dropin_latency.txt | tail -8 >> $OUT
cat $OUT
