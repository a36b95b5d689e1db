#!/bin/bash
# A/B of library builds on workloads, alternating: bash tools/ab_libs.sh OUT STEPS "wl1 wl2" label:V2:lib ...
OUT=$1; STEPS=$2; WLS=$3; shift 3
: > $OUT
for rep in 1 2 3; do
  for wl in $WLS; do
    for spec in "$@"; do
      IFS=: read -r label v2 lib <<< "$spec"
      RTLWS_V2=$v2 RTLWS_HIP_LIB=$lib python3 bench.py --workload $wl --steps $STEPS --no-cpu-baseline 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %-12s frac %.4f us %.2f parity %s' % ('$wl', '$label', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'FAILED' if d['parity'].get('failed') else 'ok'))" >> $OUT || echo "$wl $label FAILED" >> $OUT
    done
  done
done
cat $OUT
