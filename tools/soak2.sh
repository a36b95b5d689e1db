#!/bin/bash
# Extra randomised differential soak: the fuzzers with the v2 kernels forced on and off
OUT=gpurun_out/r03_fuzz_soak2.txt
: > $OUT
for s in 404 405 406; do echo "RTLWS_V2=1:" >> $OUT; RTLWS_V2=1 timeout -k 10 200 python tests/tools/fuzz_parity.py $s 50 2>&1 | tail -1 >> $OUT; done
for s in 407 408 409; do echo "RTLWS_V2=0:" >> $OUT; RTLWS_V2=0 timeout -k 10 200 python tests/tools/fuzz_parity.py $s 50 2>&1 | tail -1 >> $OUT; done
for s in 33 34; do timeout -k 10 200 python tests/tools/fuzz_parity_f64.py $s 50 2>&1 | tail -1 >> $OUT; done
echo "RTLWS_F64_FUSED=0:" >> $OUT; RTLWS_F64_FUSED=0 timeout -k 10 200 python tests/tools/fuzz_parity_f64.py 35 40 2>&1 | tail -1 >> $OUT
cat $OUT
