#!/bin/bash
# longer randomised soak of the round's new paths (f64 kernels, f32 rows, input kinds), new seeds
OUT=gpurun_out/r04_long_soak.txt; : > $OUT
for s in 51 52 53 54 55 56; do timeout -k 10 200 python tests/tools/fuzz_parity_f64.py $s 90 2>&1 | tail -1 >> $OUT; echo "." ; done
for s in 421 422 423; do timeout -k 10 200 python tests/tools/fuzz_parity.py $s 90 2>&1 | tail -1 >> $OUT; echo "."; done
cat $OUT
