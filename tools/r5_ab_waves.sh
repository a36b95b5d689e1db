#!/bin/bash
# A/B of the one-transposition f64 kernel: one-wavefront workgroups (rows dealt statically) against
# eight-wavefront workgroups whose wavefronts take rows from an LDS counter; alternating, same call
set -o pipefail
OUT=gpurun_out/r05_ab_waves.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
timeout -k 10 600 python3 -m pytest tests/test_f64_1024x_gpu.py tests/test_f64_fused_r4_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
for rep in 1 2 3; do for w in 1 8; do for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64; do
RTLWS_F64_X_WAVES=$w timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s waves %s us %.2f frac %.4f sclk %.3f GHz  -> %.1f k shader cycles per launch; parity %s' % ('$wl', '$w', r['avg_launch_us'], r['frac'], r['sclk_ghz'], r['avg_launch_us']*r['sclk_ghz'], 'FAILED' if d['parity'].get('failed') else 'ok'))" >> $OUT || echo "waves $w FAILED" >> $OUT
done; done; done
RTLWS_HIP_LIB=$V/x_stamp/librtlws_hip.so timeout -k 10 300 python3 tools/r5_wave_timeline.py w8 >> $OUT 2>&1 || echo "timeline FAILED" >> $OUT
cat $OUT
