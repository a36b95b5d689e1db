#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_f64_1024x_gpu.py tests/test_f64_fused_r4_gpu.py -x -q > gpurun_out/r04_x1024b.pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04_x1024b.pytest.log
OUT=gpurun_out/r04_x1024_store_ab.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
run() { RTLWS_F64_X1024=$4 RTLWS_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --workload $3 --steps 1500 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-38s %-12s frac %.4f us %.2f parity %s' % ('$3', '$1', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'FAILED' if d['parity'].get('failed') else 'ok'))" >> $OUT || echo "$3 $1 FAILED" >> $OUT; }
for rep in 1 2 3; do
  wl=batched_1024pt_64k_frames_f64c_f32o
  run staged "" $wl 1
  run direct $V/x_direct/librtlws_hip.so $wl 1
  run old_kernel "" $wl 0
  [ $rep = 1 ] && run nostore $V/x_nostore/librtlws_hip.so $wl 1
done
cat $OUT
