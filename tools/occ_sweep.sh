#!/bin/bash
# occupancy sweep of the ablation builds: resident wavefronts per CU via RTLWS_BLOCKS_PER_CU
for lib in variants/abl_nomem_nolds variants/abl_nomem . ; do
  for b in 4 8 12 16; do
    RTLWS_BLOCKS_PER_CU=$b RTLWS_HIP_LIB=$PWD/rtl-ws_amd/lib/$lib/librtlws_hip.so python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s waves/CU %2d  %8.2f us' % ('$lib', $b, d['roofline']['avg_launch_us']))"
  done
done
