#!/bin/bash
# round 6 diagnostics of configs[2] in f64 (hann_4096pt_k8_db_f64c_f32o): what a frame waits for
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r06_f64_4096_diag.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
WL=hann_4096pt_k8_db_f64c_f32o
for rep in 1 2; do
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="1-wg-per-CU" RTLWS_F64_BLOCKS_PER_CU=1 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
for v in nobar34 nobar nolds noload; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/f6_$v/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done
R5_LABEL="nobar,1-wg" RTLWS_F64_BLOCKS_PER_CU=1 RTLWS_HIP_LIB=$V/f6_nobar/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo "FAILED" >> $OUT
done
cat $OUT
for wl in hann_4096pt_k8_db_f64c_f32o hann_4096pt_k8_db cic8_2048pt_f64; do
  bash tools/profile_gpu.sh r06_$wl $wl > gpurun_out/prof_r06_$wl.log 2>&1
  echo "profiled $wl"; ls gpurun_out/prof_r06_$wl | head -20
done
