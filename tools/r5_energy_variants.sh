#!/bin/bash
# free re-spellings of spectrum_f64_1024x under the energy accumulator (mJ per launch, 15 000 launches, alternating):
# the sixteen 256-byte pieces of a row stored in address order; pass B as multiply-then-butterfly.  Builds:
#   make -C rtl-ws_amd xvariant NAME=xe_fft16plain EXTRA=-DRTLWS_FFT16_FMA=0
#   xe_ascst: the store loop of the kernel's epilogue run over uu = 0 .. 15 with u = rev16(uu ^ 8) (piece uu is held in
#   slot rev16(uu ^ 8): rev16 is an involution) -- a six-line #ifdef that was removed again after this measurement
#   (no difference: hipcc orders the stores by when their values are ready; profiles/r05_energy_variants_not_adopted.txt)
OUT=gpurun_out/r05_energy_variants.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2 3; do
R5_LABEL="product" timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo FAILED >> $OUT
for v in ascst fft16plain; do
R5_LABEL="$v" RTLWS_HIP_LIB=$V/xe_$v/librtlws_hip.so timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT || echo "$v FAILED" >> $OUT
done; done
RTLWS_HIP_LIB=$V/xe_fft16plain/librtlws_hip.so timeout -k 10 300 python3 -m pytest tests/test_f64_1024x_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $OUT
RTLWS_HIP_LIB=$V/xe_ascst/librtlws_hip.so timeout -k 10 300 python3 -m pytest tests/test_f64_1024x_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $OUT
cat $OUT
