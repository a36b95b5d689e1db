#!/bin/bash
# build.sh OUTDIR SANITIZER  -- the product's host sources + the fake shim + the oracle, instrumented, into OUTDIR/host_stress
# SANITIZER: thread | address,undefined | none.  Test infrastructure: nothing here is installed or shipped.
set -e
OUT=$1; SAN=$2
cd "$(dirname "$0")/../.."
mkdir -p $OUT
FLAGS="-O1 -g -fno-omit-frame-pointer"
[ "$SAN" != "none" ] && FLAGS="$FLAGS -fsanitize=$SAN"
INC="-Iinclude -Irtl-ws_amd/csrc -Irtl-ws_amd/host -Ioracle"
OBJS=""
for s in host_ctx spectrum_gpu rf_decimator_gpu stream_gpu multi_batch topology cbb_gpu synth_sensor synth_signal_source; do
  gcc $FLAGS -fPIC -Wall -Wextra -std=gnu99 $INC -c rtl-ws_amd/host/$s.c -o $OUT/$s.o
  OBJS="$OBJS $OUT/$s.o"
done
gcc $FLAGS -fPIC -Wall -Wextra -Wno-unused-parameter -std=gnu99 -ffp-contract=off $INC -c oracle/rtlws_oracle.c -o $OUT/oracle.o
g++ $FLAGS -fPIC -Wall -std=c++17 $INC -c tests/fake_hip/fake_rtlws_hip.cpp -o $OUT/fake.o
gcc $FLAGS -Wall -Wextra -std=gnu99 $INC -c tests/fake_hip/host_stress.c -o $OUT/stress.o
g++ $FLAGS -o $OUT/host_stress $OUT/stress.o $OBJS $OUT/oracle.o $OUT/fake.o -lpthread -lm
echo built $OUT/host_stress
