#!/bin/bash
# the clock probe (one sleeping wavefront on a queue of its own) beside kernels that fill every register of a SIMD:
# with / without it, every default-line workload; then the 4096-point f64 forms without it
OUT=gpurun_out/r06_clock_probe_perturbation.txt; : > $OUT
for wl in hann_4096pt_k8_db_f64c_f32o hann_4096pt_k8_db cic8_2048pt cic8_2048pt_f64 cic12_2048pt batched_1024pt_64k_frames batched_1024pt_64k_frames_f64 batched_1024pt_64k_frames_f64c_f32o rect_4096pt_f64 rect_2048pt_f64 rect_4096pt hann_4096pt_k1_db; do
R5_LABEL="probe" R6_PROBE=1 timeout -k 10 120 python3 tools/energy_per_launch.py $wl 2000 2>/dev/null >> $OUT || echo "$wl FAILED" >> $OUT
R5_LABEL="no probe" timeout -k 10 120 python3 tools/energy_per_launch.py $wl 2000 2>/dev/null >> $OUT || echo "$wl FAILED" >> $OUT
done
cat $OUT
OUT=gpurun_out/r06_ab_4096_f64_forms.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
WL=hann_4096pt_k8_db_f64c_f32o
for rep in 1 2 3; do
R5_LABEL="r05 (FLAT loads)" RTLWS_F64_Y4096=0 RTLWS_HIP_LIB=$V/r05/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="pairs from LDS" RTLWS_F64_Y4096=0 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="y: one cross exchange" RTLWS_F64_Y4096=1 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="z: two teams" RTLWS_F64_Y4096=2 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done
R5_LABEL="pairs from LDS, 1 wg/CU" RTLWS_F64_BLOCKS_PER_CU=1 RTLWS_F64_Y4096=0 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT
R5_LABEL="y, 1 wg/CU" RTLWS_F64_BLOCKS_PER_CU=1 RTLWS_F64_Y4096=1 timeout -k 10 120 python3 tools/energy_per_launch.py $WL 3000 2>/dev/null >> $OUT
cat $OUT
