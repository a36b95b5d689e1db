#!/bin/bash
# A/B: Hann inside the first butterfly layer of pass 1 (integer sums and differences of the bytes, one FMA per
# component) against the per-sample weights of the product before it; parity first, then three alternations.
set -o pipefail
OUT=gpurun_out/r06_ab_hann_in_first_butterfly.txt; : > $OUT
timeout -k 10 900 python3 -m pytest tests/test_f64_fused_gpu.py tests/test_f64_fused_r4_gpu.py tests/test_f64_4096_rows_gpu.py tests/test_fullsize_gpu.py tests/test_spectrum_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT || exit 1
V=$PWD/rtl-ws_amd/lib/variants
for rep in 1 2 3; do for wl in hann_4096pt_k8_db_f64c_f32o hann_4096pt_k8_db_f64; do
R5_LABEL="per-sample weights (before)" RTLWS_HIP_LIB=$V/before_hann_bfly/librtlws_hip.so timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
R5_LABEL="weights in the first butterfly layer" timeout -k 10 120 python3 tools/energy_per_launch.py $wl 3000 2>/dev/null >> $OUT || echo FAILED >> $OUT
done; done
grep -v amdgpu.ids $OUT
