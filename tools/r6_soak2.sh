#!/bin/bash
# after the Hann change: fresh fuzz seeds on the final tree (the f64 fuzzer draws windows, sizes, K and output modes)
OUT=gpurun_out/r06_soak_final_tree.txt; : > $OUT
for s in 81 82 83 84 85 86; do timeout -k 10 200 python3 tests/tools/fuzz_parity_f64.py $s 60 2>&1 | tail -1 >> $OUT || exit 1; done
for s in 631; do timeout -k 10 200 python3 tests/tools/fuzz_parity.py $s 60 2>&1 | tail -1 >> $OUT || exit 1; done
cat $OUT
