#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_y4096_phase_times.txt; : > $OUT
L=$PWD/rtl-ws_amd/lib/variants/ys255/librtlws_hip.so
for sets in 1 4; do
R6_SETS=$sets RTLWS_F64_Y4096=1 RTLWS_HIP_LIB=$L timeout -k 10 120 python3 tools/r6_phase_times.py >> $OUT 2>&1 || echo FAILED >> $OUT
R6_SETS=$sets RTLWS_F64_Y4096=1 RTLWS_HIP_LIB=$L RTLWS_F64_BLOCKS_PER_CU=1 timeout -k 10 120 python3 tools/r6_phase_times.py >> $OUT 2>&1 || echo FAILED >> $OUT
done
grep -v amdgpu.ids $OUT
