#!/bin/bash
OUT=gpurun_out/r06_wave_timeline_f64_fused.txt; : > $OUT
RTLWS_HIP_LIB=$PWD/rtl-ws_amd/lib/variants/f_stamp/librtlws_hip.so timeout -k 10 200 python3 tools/r6_wave_timeline.py cic8_2048pt_f64 hann_4096pt_k8_db_f64c_f32o rect_2048pt_f64 >> $OUT 2>&1 || echo FAILED >> $OUT
grep -v amdgpu.ids $OUT
