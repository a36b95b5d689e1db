#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r06_y4096_blocky.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
WL=hann_4096pt_k8_db_f64c_f32o
for rep in 1 2; do
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
for m in 15 1 2 4 8; do
R5_LABEL="pins=$m" RTLWS_HIP_LIB=$V/y_blocky$m/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo "$m FAILED" >> $OUT
done
R5_LABEL="stamp-build" RTLWS_HIP_LIB=$V/y_stamp/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo "FAILED" >> $OUT
done
cat $OUT
bash tools/r6_phase.sh
