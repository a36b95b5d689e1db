#!/bin/bash
# round 4: whole GPU suite, then the drivers' lines (multi-stream latency after the warm start, multi-batch)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_suite.pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r04_suite.pytest.log
tail -15 gpurun_out/r04_suite.pytest.log
M=rtl-ws_amd/lib/rtlws_multi_stream
: > gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 5 --rate 2400000 --output payload >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 5 --rate 2400000 --output f32 >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 5 --rate 2400000 --output f32 --precision f64 >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 5 --rate 2400000 --output f32 --precision f64c_f32o >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 1 --seconds 4 --unpaced --output f32 >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 1 --seconds 4 --unpaced --output f32 --precision f64 >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 4 --unpaced --output f32 >> gpurun_out/r04_multi_stream.jsonl
timeout -k 10 60 $M --streams 8 --seconds 4 --unpaced --output payload --chunk-buffers 4 >> gpurun_out/r04_multi_stream.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r04_multi_stream.jsonl"):
    r=json.loads(l); print(r["streams"], r["paced"], r["output"], r.get("precision"), "total %.4g drops %d lat avg %.3f max %.3f" % (r["spectra_per_s_total"], r["chunks_dropped"], r["latency_ms_avg"], r["latency_ms_max"]))
PY
B=rtl-ws_amd/lib/rtlws_multi_batch
: > gpurun_out/r04_multi_batch.jsonl
timeout -k 10 100 $B --frames 65536 --launches 2000 >> gpurun_out/r04_multi_batch.jsonl
timeout -k 10 100 $B --frames 65536 --launches 1000 --precision f64c_f32o >> gpurun_out/r04_multi_batch.jsonl
timeout -k 10 100 $B --frames 65536 --launches 1000 --shards-on-device0 2 >> gpurun_out/r04_multi_batch.jsonl
cat gpurun_out/r04_multi_batch.jsonl | cut -c1-600
timeout -k 10 100 python tools/dropin_latency.py > gpurun_out/r04_dropin_latency.txt 2>&1; tail -12 gpurun_out/r04_dropin_latency.txt
