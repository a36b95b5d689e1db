#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_f64_1024x_gpu.py tests/test_f64_fused_gpu.py tests/test_f64_fused_r4_gpu.py -x -q > gpurun_out/r04_x1024.pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r04_x1024.pytest.log
OUT=gpurun_out/r04_ab_x1024.txt; : > $OUT
run() { RTLWS_F64_X1024=$1 timeout -k 10 120 python3 bench.py --workload $2 --steps 1500 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-38s x1024=%s frac %.4f us %.2f strict %.3g parity %s' % ('$2', '$1', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['parity'].get('max_rel_err_floor1e-9', -1), 'FAILED' if d['parity'].get('failed') else 'ok'))" >> $OUT || echo "$2 $1 FAILED" >> $OUT; }
for rep in 1 2 3; do for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64; do run 1 $wl; run 0 $wl; done; done
cat $OUT
