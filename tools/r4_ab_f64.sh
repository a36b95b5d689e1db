#!/bin/bash
# A/B of f64 fused builds (lib/variants/*) on the f64-arithmetic workloads, alternating, 3 rounds
OUT=${1:-gpurun_out/r04_ab_f64.txt}; STEPS=${2:-1500}
V=$PWD/rtl-ws_amd/lib/variants
mkdir -p gpurun_out
: > $OUT
run() { # label lib blocks workload
  RTLWS_F64_BLOCKS_PER_CU=$3 RTLWS_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --workload $4 --steps $STEPS --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-38s %-14s frac %.4f us %.2f parity %s' % ('$4', '$1', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'FAILED' if d['parity'].get('failed') else 'ok'))" >> $OUT || echo "$4 $1 FAILED" >> $OUT
}
for rep in 1 2 3; do
  for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64; do
    run product "" 0 $wl
    run noint $V/f64_noint/librtlws_hip.so 0 $wl
    run w3_12 $V/f64_w3/librtlws_hip.so 12 $wl
    run w3_8 $V/f64_w3/librtlws_hip.so 8 $wl
    if [ $rep = 1 ]; then
      run nolds $V/f64_nolds/librtlws_hip.so 0 $wl
      run nostore $V/f64_nostore/librtlws_hip.so 0 $wl
      run noload $V/f64_noload/librtlws_hip.so 0 $wl
    fi
  done
done
cat $OUT
