#!/bin/bash
# Sanitizer pass on the CPU-side C code (GPU AddressSanitizer is not available on this
# pool): the oracle is rebuilt with ASan+UBSan and the CPU test-suite is run against it.
set -e
cd "$(dirname "$0")/../.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -ffp-contract=off -shared \
    -o /tmp/librtlws_oracle_asan.so oracle/rtlws_oracle.c -lm -lpthread
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0 RTLWS_ORACLE_LIB=/tmp/librtlws_oracle_asan.so \
    python -m pytest tests/test_oracle_cpu.py -q -x
