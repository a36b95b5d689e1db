#!/bin/bash
# Round-5 evidence run (one gpurun call): the driver's default line twice (default steps, --steps 20), the rocprofv3
# kernel-trace + PMC passes of the three configs[1] workloads (own output directory each), the drop-in latencies, the
# fuzz soak.  (The whole GPU suite and the first profile of the headline: tools/r5_suite.sh.)
TAG=r05
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default_line.err
echo "default line rc=$?"
python3 bench.py --steps 20 > gpurun_out/${TAG}_bench_default_line_steps20.json 2>> gpurun_out/${TAG}_bench_default_line.err
echo "steps20 line rc=$?"
for wl in batched_1024pt_64k_frames batched_1024pt_64k_frames_f64 batched_1024pt_64k_frames_f64c_f32o; do
  timeout -k 10 500 bash tools/profile_gpu.sh ${TAG}_$wl $wl > gpurun_out/prof_${TAG}_$wl.log 2>&1
  echo "profiled $wl"
done
timeout -k 10 200 python3 tools/dropin_latency.py > gpurun_out/${TAG}_dropin_latency.txt 2>&1
cat gpurun_out/${TAG}_dropin_latency.txt
OUT=gpurun_out/${TAG}_fuzz_soak.txt; : > $OUT
for s in 511 512; do timeout -k 10 200 python3 tests/tools/fuzz_parity.py $s 45 2>&1 | tail -1 >> $OUT; done
for s in 51 52 53; do timeout -k 10 200 python3 tests/tools/fuzz_parity_f64.py $s 45 2>&1 | tail -1 >> $OUT; done
timeout -k 10 300 python3 tests/tools/big_batch_check.py 2>&1 | tail -2 >> $OUT
python3 -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
cat $OUT
