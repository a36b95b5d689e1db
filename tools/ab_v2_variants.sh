#!/bin/bash
# One call: the 4096-point workloads on the four-wavefront kernel, the product v2 kernel and v2 build variants.
OUT=${1:-gpurun_out/r03_ab_v2_variants.txt}
STEPS=${2:-1200}
: > $OUT
run() {  # label, V2 flag, lib ("" = product), workload
  RTLWS_V2=$2 RTLWS_HIP_LIB=$3 python3 bench.py --workload $4 --steps $STEPS --no-cpu-baseline 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %-12s frac %.4f us %.2f' % ('$4', '$1', d['roofline']['frac'], d['roofline']['avg_launch_us']))" >> $OUT || echo "$4 $1 FAILED" >> $OUT
}
V=rtl-ws_amd/lib/variants
for rep in 1 2; do
  for wl in hann_4096pt_k8_db hann_4096pt_k1_db rect_4096pt_k8 rect_4096pt; do
    run v1 0 "" $wl
    run v2 1 "" $wl
    run v2_winregs 1 $V/v2_winregs/librtlws_hip.so $wl
    run v2_nopf 1 $V/v2_nopf/librtlws_hip.so $wl
    run v2_nt 1 $V/v2_nt/librtlws_hip.so $wl
  done
done
cat $OUT
