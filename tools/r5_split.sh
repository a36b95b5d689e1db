#!/bin/bash
# engine option "split" and larger launches on configs[1]: tests, then alternating A/B (same call)
set -o pipefail
OUT=gpurun_out/r05_split.txt; : > $OUT
timeout -k 10 900 python3 -m pytest tests/test_split_gpu.py tests/test_bench_gpu.py tests/test_graph_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $OUT || { cat $OUT; exit 1; }
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r.get('energy') or {}; print('%-40s %-14s us %.2f frac %.4f frac_wall %.4f sclk %.3f GHz; %.1f mJ per launch at %.0f W; parity %s' % (d['config']['workload'], '$1', r['avg_launch_us'], r['frac'], r['frac_wall'], r.get('sclk_ghz', 0), e.get('mj_per_launch', 0), e.get('watts', 0), 'FAILED' if d['parity'].get('failed') else 'ok'))"; }
for rep in 1 2 3; do for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames; do
for q in 1 2 3 4; do
timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra --split $q 2>/dev/null | line "split=$q" >> $OUT || echo "$wl split $q FAILED" >> $OUT
done
timeout -k 10 120 python3 bench.py --workload $wl --steps 500 --no-cpu-baseline --no-extra --frames 262144 2>/dev/null | line "262144 frames" >> $OUT || echo "$wl 262144 FAILED" >> $OUT
timeout -k 10 120 python3 bench.py --workload $wl --steps 500 --no-cpu-baseline --no-extra --frames 262144 --split 2 2>/dev/null | line "262144 split=2" >> $OUT || echo "$wl 262144 split FAILED" >> $OUT
done; done
cat $OUT
timeout -k 10 300 python3 bench.py > gpurun_out/r05_bench_default_line_a.json 2> gpurun_out/r05_bench_default_line_a.err; echo "default line rc=$?"; tail -c 600 gpurun_out/r05_bench_default_line_a.err
