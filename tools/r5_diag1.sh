#!/bin/bash
# round 5, first diagnostic call: what the two wavefronts of a SIMD do to each other in spectrum_f64_1024x.hip
# (build first: make -C rtl-ws_amd variant NAME=x_stamp EXTRA=-DRTLWS_X_STAMP; hipcc tools/valubench.hip)
set -o pipefail
OUT=gpurun_out/r05_diag1.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
echo "== valubench, f64 classes: ns per instruction and SIMD at 1/2/3/4/8 wavefronts per SIMD" >> $OUT
VALUBENCH_ONLY=f64 timeout -k 10 200 tools/build/valubench >> $OUT 2>&1 || echo "valubench FAILED" >> $OUT
echo "== per-wavefront timeline, spectra_f64_1024x f32 rows" >> $OUT
RTLWS_HIP_LIB=$V/x_stamp/librtlws_hip.so timeout -k 10 300 python3 tools/r5_wave_timeline.py 8 4 6 >> $OUT 2>&1 || echo "timeline FAILED" >> $OUT
echo "== workgroups per CU (one wavefront each): us per launch, clock, cycles" >> $OUT
for rep in 1 2; do for b in 8 4 6 9; do
RTLWS_F64_BLOCKS_PER_CU=$b timeout -k 10 120 python3 bench.py --workload batched_1024pt_64k_frames_f64c_f32o --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('per_cu %-3s us %.2f frac %.4f sclk %.3f GHz  -> %.1f k shader cycles per launch' % ('$b', r['avg_launch_us'], r['frac'], r['sclk_ghz'], r['avg_launch_us']*r['sclk_ghz']))" >> $OUT || echo "per_cu $b FAILED" >> $OUT
done; done
cat $OUT
