#!/bin/bash
# A/B of the two 4096-point kernels in one call: RTLWS_V2=0 (four wavefronts per frame) against
# RTLWS_V2=1 (two virtual threads per lane), alternating.  usage: bash tools/ab_v2.sh [out] [steps]
OUT=${1:-gpurun_out/r03_ab_v2.txt}
STEPS=${2:-1500}
: > $OUT
for rep in 1 2 3; do
  for wl in hann_4096pt_k8_db rect_4096pt; do
    for v in 0 1; do
      RTLWS_V2=$v python3 bench.py --workload $wl --steps $STEPS --no-cpu-baseline 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl V2=$v frac %.4f wall %.4f us %.2f value %.4g parity %s' % (d['roofline']['frac'], d['roofline']['frac_wall'], d['roofline']['avg_launch_us'], d['value'], json.dumps(d['parity'])))" >> $OUT || exit 1
    done
  done
done
cat $OUT
