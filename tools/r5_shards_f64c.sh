#!/bin/bash
# the f64-arithmetic / f32-row kernel with 1 / 2 / 3 concurrent shards of ONE device's batch on
# independent queues (rtlws_multi_batch --shards-per-device Q: own engine, queue and host thread per shard, no event
# between them), 65 536 and 262 144 frames, three alternations; wall-clock fractions (the shards' launches overlap)
OUT=gpurun_out/r05_shards_one_device_f64c.txt; : > $OUT
L=rtl-ws_amd/lib/rtlws_multi_batch
for rep in 1 2 3; do for frames in 65536 262144; do for q in 1 2 3; do
  n=$((frames == 65536 ? 1500 : 400))
  timeout -k 10 120 $L --precision f64c_f32o --frames $frames --launches $n --warmup 500 --shards-per-device $q 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['algorithmic_bytes_per_frame']*d['frames_used']; w=d['wall_ms']/d['launches']
print('f64c_f32o frames %7d shards %d: %.2f us per batch (wall), %.4f of the HBM roofline, %.4g spectra/s; per-shard event us %s' % (d['frames_used'], d['shards'], 1e3*w, b/(w*1e-3)/8e12, d['spectra_per_s_total'], ' '.join('%.1f' % (1e3*s['event_ms_per_launch']) for s in d['per_shard'])))" >> $OUT || echo "frames $frames q $q FAILED" >> $OUT
done; done; done
cat $OUT
