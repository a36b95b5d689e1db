#!/bin/bash
# configs[2] in the reference's arithmetic (hann_4096pt_k8_db_f64c_f32o): the pass-3 pairs from LDS instead of FLAT
# loads -- parity tests, then same-call A/B against the previous build and the ablation builds, in joules
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_f64_fused_gpu.py tests/test_f64_fused_r4_gpu.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r06_hann64_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r06_hann64_tests.txt
[ $rc -eq 0 ] || exit $rc
OUT=gpurun_out/r06_f64_4096_tw3_from_lds.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2 3; do
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py hann_4096pt_k8_db_f64c_f32o 4000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="r05-build" RTLWS_HIP_LIB=$V/f6_old/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py hann_4096pt_k8_db_f64c_f32o 4000 2>/dev/null >> $OUT || echo "old FAILED" >> $OUT
done
for rep in 1 2; do
for v in nolds noload; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/f6_$v/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py hann_4096pt_k8_db_f64c_f32o 4000 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py hann_4096pt_k8_db_f64 4000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py rect_4096pt_k8 4000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done
cat $OUT
