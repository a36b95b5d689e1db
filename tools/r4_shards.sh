#!/bin/bash
# throughput of ONE device when a batch is cut into S concurrent shards (own engine, queue, host thread each)
B=rtl-ws_amd/lib/rtlws_multi_batch
OUT=gpurun_out/r04_shards_one_device.txt; : > $OUT
for rep in 1 2; do
for S in 1 2 3 4 8; do
  for F in 65536 262144; do
    timeout -k 10 100 $B --frames $F --launches $((2000*65536/F)) --warmup 600 --shards-on-device0 $S | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('shards %d frames %7d: total %.4g spectra/s = %.4f of 8 TB/s; per-shard event ms/launch %s' % (r['shards'], r['frames_used'], r['spectra_per_s_total'], r['spectra_per_s_total']*6144/8e12, ['%.4f' % s['event_ms_per_launch'] for s in r['per_shard']]))" >> $OUT
  done
done
done
cat $OUT
