#!/bin/bash
# Sample clocks and power (rocm-smi) while bench.py runs a long sustained loop.
# usage (via gpurun): bash tools/power_probe.sh [workload] [steps]
WL=${1:-batched_1024pt_64k_frames}
STEPS=${2:-60000}
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -20
python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --workload $WL > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 1.5
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power (W)" | sed -e 's/.*: \(.\)clk clock level: .: (\(.*\))/\1clk \2/' -e 's/.*Power (W): \(.*\)/power \1 W/' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('avg_launch_us', d['roofline']['avg_launch_us'], 'steps', d['steps'])"
