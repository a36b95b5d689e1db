#!/bin/bash
# round 5: the whole GPU suite, then the rocprofv3 evidence of the headline (f64 arithmetic, f32 rows), of the
# f32 fast mode and of the f64-row workload, then the C drivers
set -o pipefail
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest.txt 2>&1; rc=$?
tail -5 gpurun_out/r05_gputest.txt
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames batched_1024pt_64k_frames_f64; do
  timeout -k 10 600 bash tools/profile_gpu.sh r05 $wl > gpurun_out/r05_profile_$wl.log 2>&1 || echo "profile $wl FAILED"
done
ls gpurun_out/prof_r05/ | head -40
timeout -k 10 200 python3 bench.py --workload multi_batch --steps 500 > gpurun_out/r05_multi_batch.jsonl 2> gpurun_out/r05_multi_batch.err || echo "multi_batch FAILED"
timeout -k 10 200 python3 bench.py --workload multi_batch --steps 500 --shards-per-device 2 >> gpurun_out/r05_multi_batch.jsonl 2>> gpurun_out/r05_multi_batch.err || echo "multi_batch 2 FAILED"
timeout -k 10 200 python3 bench.py --workload realtime_8x2400k --steps 4000 > gpurun_out/r05_multi_stream.jsonl 2> gpurun_out/r05_multi_stream.err || echo "realtime FAILED"
tail -c 1500 gpurun_out/r05_multi_batch.jsonl; tail -c 800 gpurun_out/r05_multi_stream.jsonl
