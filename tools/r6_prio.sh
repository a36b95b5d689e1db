#!/bin/bash
# experiment: the two wavefronts of a SIMD alternate at the higher issue priority, row by row (s_setprio), in
# spectrum_f64_fused.hip -- does the younger one stop finishing alone?  (cic8_2048pt_f64 is under the power cap)
OUT=gpurun_out/r06_ab_setprio_rows.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2 3; do for wl in cic8_2048pt_f64 rect_2048pt_f64 hann_4096pt_k8_db_f64c_f32o; do
R5_LABEL="product" timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="prio 0/1" RTLWS_HIP_LIB=$V/f_prio1/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="prio 0/3" RTLWS_HIP_LIB=$V/f_prio3/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done; done
RTLWS_HIP_LIB=$V/f_prio1s/librtlws_hip.so timeout -k 10 200 python3 tools/r6_wave_timeline.py cic8_2048pt_f64 >> $OUT 2>&1
grep -v amdgpu.ids $OUT
