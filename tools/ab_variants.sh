#!/bin/bash
# A/B experiment runner for the GPU box: bench every library under
# rtl-ws_amd/lib/variants/*/ (plus the product build) on the same workload.
# usage (via gpurun): bash tools/ab_variants.sh [workload] [rounds]
WL=${1:-batched_1024pt_64k_frames}
ROUNDS=${2:-2}
OUT=gpurun_out/ab_$(date +%H%M%S).txt
for r in $(seq 1 $ROUNDS); do
  for lib in rtl-ws_amd/lib/librtlws_hip.so rtl-ws_amd/lib/variants/*/librtlws_hip.so; do
    [ -f "$lib" ] || continue
    RTLWS_HIP_LIB=$PWD/$lib python3 bench.py --steps 400 --warmup 30 --no-cpu-baseline --workload $WL 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-70s %8.2f us  %6.0f GB/s  %.3e /s' % ('$lib'.replace('rtl-ws_amd/lib/',''), d['roofline']['avg_launch_us'], d['roofline']['achieved'], d['value']))" | tee -a $OUT
  done
done
