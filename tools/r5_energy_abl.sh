#!/bin/bash
# where the joules of one launch of spectrum_f64_1024x (f32 rows, 65 536 frames) go: energy-attribution
# builds (each removes one part, results wrong) under the package energy accumulator
set -o pipefail
OUT=gpurun_out/r05_energy_ablations.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2; do
R5_LABEL="product" timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo FAILED >> $OUT
for v in nolds nostore noload nomem nopassa notwb nopassb nofft; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/xe_$v/librtlws_hip.so timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done; done
cat $OUT
