#!/bin/bash
# the split option with its side queues at the highest stream priority (own hardware queues): standalone Q = 1..4 and
# inside the default line (where round 5's first form took 14 ms per batch)
OUT=gpurun_out/r05_split_priority_queues.txt; : > $OUT
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %-10s us %.2f frac %.4f frac_wall %.4f; parity %s' % (d['config']['workload'], '$1', r['avg_launch_us'], r['frac'], r['frac_wall'], 'FAILED' if d['parity'].get('failed') else 'ok'))"; }
for rep in 1 2; do for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames; do for q in 1 2 3 4; do
timeout -k 10 120 python3 bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-extra --no-energy --split $q 2>/dev/null | line "split=$q" >> $OUT || echo "$wl split $q FAILED" >> $OUT
done; done; done
timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
for x in d['extra_workloads']: print('default line: %-40s split %d frac %.4f frac_wall %.4f' % (x['workload'], x['split'], x['roofline']['frac'], x['roofline']['frac_wall']))" >> $OUT
timeout -k 10 300 python3 -m pytest tests/test_split_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $OUT
cat $OUT
