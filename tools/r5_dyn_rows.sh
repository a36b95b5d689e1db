#!/bin/bash
# EXPERIMENT: rows of the CIC-fused f32 kernels claimed from per-XCD device counters (-DRTLWS_DYN_ROWS variant)
# against rows dealt statically: every-row parity through the GPU tests, then three alternations
set -o pipefail
OUT=gpurun_out/r05_ab_dynamic_rows_cic.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants/dyn/librtlws_hip.so
RTLWS_HIP_LIB=$V timeout -k 10 600 python3 -m pytest tests/test_spectrum_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "cic" 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r.get('energy') or {}; print('%-16s %-10s us %.2f frac %.4f sclk %.3f GHz; %.1f mJ per launch at %.0f W; parity %s' % (d['config']['workload'], '$1', r['avg_launch_us'], r['frac'], r.get('sclk_ghz', 0), e.get('mj_per_launch', 0), e.get('watts', 0), 'FAILED' if d['parity'].get('failed') else 'ok'))"; }
for rep in 1 2 3; do for wl in cic8_2048pt cic12_2048pt; do
timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | line static >> $OUT || echo "$wl static FAILED" >> $OUT
RTLWS_HIP_LIB=$V timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | line dynamic >> $OUT || echo "$wl dynamic FAILED" >> $OUT
done; done
cat $OUT
