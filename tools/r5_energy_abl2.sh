#!/bin/bash
# second set of energy-attribution builds of spectrum_f64_1024x (the "everything else" of DESIGN.md 6.3): no integer
# radix-4, no cross-row transpose, no conversions, the three together, no |X|^2 -- mJ per launch, alternating
OUT=gpurun_out/r05_energy_ablations_front_end.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2; do
R5_LABEL="product" timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo FAILED >> $OUT
for v in nopass0 noswap nocvt nofront nopow nofft; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/xe_$v/librtlws_hip.so timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done; done
cat $OUT
