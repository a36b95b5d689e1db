#!/bin/bash
# Clock / package power / us per launch of one workload on several builds, one after the other
# (rocm-smi sampled during a long run of each).  usage: bash tools/power_ab.sh OUT WORKLOAD STEPS label:V2:lib ...
OUT=$1; WL=$2; STEPS=$3; shift 3
for spec in "$@"; do
  IFS=: read -r label v2 lib <<< "$spec"
  echo "== $WL $label (RTLWS_V2=$v2 lib=${lib:-product})" >> $OUT
  RTLWS_V2=$v2 RTLWS_HIP_LIB=$lib python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --workload $WL > /tmp/pp_bench.json 2>/dev/null &
  BP=$!
  sleep 2.5
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)" | sed -e 's/.*: sclk clock level: .: (\(.*\))/sclk \1/' -e 's/.*Power (W): \(.*\)/power \1 W/' | tr '\n' ' ' >> $OUT; echo >> $OUT
    sleep 0.4
  done
  wait $BP
  python3 -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('avg_launch_us %.2f frac %.4f steps %d' % (d['roofline']['avg_launch_us'], d['roofline']['frac'], d['steps']))" >> $OUT
done
