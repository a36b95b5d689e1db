#!/bin/bash
# usage (via gpurun): bash tools/valupower.sh   -- power and clock under scalar vs packed f32 math
for m in add pkadd fma pkfma; do
  tools/build/valupower $m 4 > /tmp/vp_$m.txt &
  P=$!
  sleep 2
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)" | tr -s ' \t' ' ' | tr '\n' ';'; echo
  sleep 1
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)" | tr -s ' \t' ' ' | tr '\n' ';'; echo
  wait $P
  cat /tmp/vp_$m.txt
done
