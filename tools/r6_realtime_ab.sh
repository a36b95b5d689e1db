#!/bin/bash
# eight paced 2.4 MS/s sensors on one device: this round's build against round 5's (git archive d535711, built beside
# it under tools/build/, not committed), alternating, three times
OUT=gpurun_out/r06_realtime_ab.txt; : > $OUT
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.1f spectra/s per stream, dropped %d, latency avg %.3f ms max %.3f ms; per stream max: %s' % (r['spectra_per_s_total']/8, r['chunks_dropped'], r['latency_ms_avg'], r['latency_ms_max'], ' '.join('%.2f' % s['latency_ms_max'] for s in r['per_stream'])))"; }
for rep in 1 2 3; do
timeout -k 10 60 rtl-ws_amd/lib/rtlws_multi_stream --streams 8 --seconds 4 --rate 2400000 --output payload --devices 1 2>/dev/null | line "round 6" >> $OUT || echo "r6 FAILED" >> $OUT
timeout -k 10 60 tools/build/r05tree/rtl-ws_amd/lib/rtlws_multi_stream --streams 8 --seconds 4 --rate 2400000 --output payload --devices 1 2>/dev/null | line "round 5" >> $OUT || echo "r5 FAILED" >> $OUT
done
cat $OUT
