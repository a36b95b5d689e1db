#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_f64_4096y_gpu.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py tests/test_f64_fused_r4_gpu.py -x -q -m gpu > gpurun_out/r06_z4096_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r06_z4096_tests.txt
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/r06_z4096_tests.txt | head -30; exit $rc; }
OUT=gpurun_out/r06_ab_z4096.txt; : > $OUT
for rep in 1 2 3; do
for wl in hann_4096pt_k8_db_f64c_f32o; do
R5_LABEL="z (two teams)" timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="y (one team)" RTLWS_F64_Y4096=1 timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="two-exchange" RTLWS_F64_Y4096=0 timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done; done
for wl in hann_4096pt_k8_db_f64 rect_4096pt_k8; do
R5_LABEL="z" timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done
cat $OUT
