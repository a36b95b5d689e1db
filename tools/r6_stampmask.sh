#!/bin/bash
OUT=gpurun_out/r06_y4096_stamp_masks.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
WL=hann_4096pt_k8_db_f64c_f32o
R5_LABEL="product y" RTLWS_F64_Y4096=1 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 2000 2>/dev/null >> $OUT
for m in "$@"; do
R5_LABEL="stamps=$m" RTLWS_F64_Y4096=1 RTLWS_HIP_LIB=$V/ys$m/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 2000 2>/dev/null >> $OUT || echo "$m FAILED" >> $OUT
done
cat $OUT
