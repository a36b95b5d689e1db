#!/bin/bash
# energy per wave64 instruction by class, and energy per launch of the configs[1] kernels
set -o pipefail
OUT=gpurun_out/r05_energy.txt; : > $OUT
timeout -k 10 300 tools/build/f64energy 2.5 >> $OUT 2>&1 || echo "f64energy FAILED" >> $OUT
for rep in 1 2; do
R5_LABEL="x_waves=1" RTLWS_F64_X_WAVES=1 timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o >> $OUT 2>&1 || echo FAILED >> $OUT
R5_LABEL="x_waves=8" RTLWS_F64_X_WAVES=8 timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o >> $OUT 2>&1 || echo FAILED >> $OUT
R5_LABEL="two transpositions" RTLWS_F64_X1024=0 timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o >> $OUT 2>&1 || echo FAILED >> $OUT
timeout -k 10 120 python3 tools/r5_energy.py f64 >> $OUT 2>&1 || echo FAILED >> $OUT
timeout -k 10 120 python3 tools/r5_energy.py f32 >> $OUT 2>&1 || echo FAILED >> $OUT
done
cat $OUT
