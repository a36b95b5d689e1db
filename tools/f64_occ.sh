#!/bin/bash
OUT=${1:-gpurun_out/r03_f64_occ.txt}
: > $OUT
for b in 4 6 8 9; do
  RTLWS_F64_BLOCKS_PER_CU=$b python3 bench.py --workload batched_1024pt_64k_frames_f64 --steps 500 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocks/CU $b frac %.4f us %.2f' % (d['roofline']['frac'], d['roofline']['avg_launch_us']))" >> $OUT
done
cat $OUT
