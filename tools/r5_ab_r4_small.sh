#!/bin/bash
# small-batch / drop-in latencies of this round's tree against round 4's (built from git archive fa312cf into _r4/,
# not committed), alternating, same call: did the one-wavefront path regress?
OUT=gpurun_out/r05_ab_r4_small_batches.txt; : > $OUT
for rep in 1 2 3; do
  echo "== round 5 tree (pass $rep)" >> $OUT; timeout -k 10 120 python3 tools/dropin_latency.py 2>/dev/null | head -4 >> $OUT
  echo "== round 4 tree (pass $rep)" >> $OUT; (cd _r4 && timeout -k 10 120 python3 tools/dropin_latency.py 2>/dev/null | head -4) >> $OUT
  echo "== round 5 tree, f64 batches of 1 / 64 / 1024 rows (pass $rep)" >> $OUT; timeout -k 10 120 python3 tools/f64_kernel_time.py 2>/dev/null >> $OUT
  echo "== round 4 tree, the same (pass $rep)" >> $OUT; (cd _r4 && timeout -k 10 120 python3 tools/f64_kernel_time.py 2>/dev/null) >> $OUT
done
cat $OUT
