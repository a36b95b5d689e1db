#!/bin/bash
# where the joules of one launch of configs[2] in f64 go (hann_4096pt_k8_db_f64c_f32o, spectrum_f64_fused.hip N = 4096):
# energy-attribution builds (each removes one part, results wrong), no probe wavefront, two alternations
OUT=gpurun_out/r06_energy_ablations_hann_4096_f64.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
WL=hann_4096pt_k8_db_f64c_f32o
for rep in 1 2 3; do
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
for v in nolds noload nofft nobar; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/f6_$v/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done; done
R5_LABEL="product, 1 wg/CU" RTLWS_F64_BLOCKS_PER_CU=1 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT
R5_LABEL="f32 kernel" timeout -k 10 120 python3 tools/energy_per_launch.py hann_4096pt_k8_db 6000 2>/dev/null >> $OUT
cat $OUT
