#!/bin/bash
# 3 wavefronts per SIMD on the one-transposition f64 kernel (WAVES = 12) against 2 (WAVES = 8) and the static form:
# parity, then three alternations with cycles, clock and joules (VERDICT r4 item 1b's measurement)
set -o pipefail
OUT=gpurun_out/r05_ab_waves12.txt; : > $OUT
timeout -k 10 600 python3 -m pytest tests/test_f64_1024x_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r.get('energy') or {}; print('%-40s %-10s us %.2f frac %.4f sclk %.3f GHz -> %.1f k shader cycles per launch; %.1f mJ per launch at %.0f W; parity %s' % (d['config']['workload'], '$1', r['avg_launch_us'], r['frac'], r.get('sclk_ghz', 0), r['avg_launch_us'] * r.get('sclk_ghz', 0), e.get('mj_per_launch', 0), e.get('watts', 0), 'FAILED' if d['parity'].get('failed') else 'ok'))"; }
for rep in 1 2 3; do for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64; do for w in 1 8 12; do
RTLWS_F64_X_WAVES=$w timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | line "waves=$w" >> $OUT || echo "$wl waves $w FAILED" >> $OUT
done; done; done
for w in 8 12; do R5_LABEL="x_waves=$w" RTLWS_F64_X_WAVES=$w timeout -k 10 120 python3 tools/r5_energy.py f64c_f32o 2>/dev/null >> $OUT; done
cat $OUT
