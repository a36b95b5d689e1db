#!/bin/bash
# ablations of spectrum_f64_1024x.hip on the f64-arithmetic / f32-row workload + its counters.
# Build the variants first (here, before gpurun):
#   for v in NOLDS NOSTORE NOLOAD; do make -C rtl-ws_amd variant NAME=x_$(echo $v | tr A-Z a-z) EXTRA=-DRTLWS_F64_ABL_$v; done
#   make -C rtl-ws_amd variant NAME=x_nomem EXTRA="-DRTLWS_F64_ABL_NOLOAD -DRTLWS_F64_ABL_NOSTORE"
OUT=gpurun_out/r04_x1024_ablations.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
run() { RTLWS_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --workload $3 --steps 1500 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-38s %-10s frac %.4f us %.2f' % ('$3', '$1', d['roofline']['frac'], d['roofline']['avg_launch_us']))" >> $OUT || echo "$3 $1 FAILED" >> $OUT; }
for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64; do
  run product "" $wl
  for v in x_nolds x_nostore x_noload x_nomem; do run $v $V/$v/librtlws_hip.so $wl; done
done
cat $OUT
bash tools/profile_gpu.sh r04x batched_1024pt_64k_frames_f64c_f32o > /dev/null 2>&1
cat gpurun_out/prof_r04x/summary_batched_1024pt_64k_frames_f64c_f32o_pmc.json | python3 -c "
import json,sys; d=json.load(sys.stdin); print(d['dispatch']); [print(k, '%.4g' % v['mean_per_launch']) for k,v in sorted(d['counters'].items())]"
