#!/bin/bash
# Sanitizer pass on the C HOST LAYER (librtlws_amd: spectrum.h / resample.h / rf_decimator.h / stream / multi /
# audio over the shim) on the CPU: built with ASan + UBSan and driven through the CPU tests, which on a box
# without a GPU walk every failure path (no device: partially built handles torn down, void entry points
# recording their failure, the partition arithmetic).  GPU AddressSanitizer is not available on this pool.
set -e
cd "$(dirname "$0")/../.."
OUT=/tmp/rtlws_asan_host
mkdir -p $OUT
SRCS="host_ctx spectrum_gpu rf_decimator_gpu stream_gpu audio_gpu multi_batch topology"
OBJS=""
for s in $SRCS; do
  gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -Wall -Wextra -std=gnu99 \
      -Iinclude -Irtl-ws_amd/csrc -Irtl-ws_amd/host -c rtl-ws_amd/host/$s.c -o $OUT/$s.o
  OBJS="$OBJS $OUT/$s.o"
done
gcc -shared -fPIC -fsanitize=address,undefined -o $OUT/librtlws_amd.so $OBJS -Lrtl-ws_amd/lib -lrtlws_hip \
    -Wl,-rpath,$PWD/rtl-ws_amd/lib -lpthread -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 RTLWS_AMD_LIB=$OUT/librtlws_amd.so \
    python -m pytest tests/test_abi_cpu.py tests/test_multi_batch_cpu.py tests/test_topo_cpu.py -q -x \
    -k "not call_graph and not launch_path and not cbb_init and not exported"
