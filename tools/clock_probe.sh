#!/bin/bash
# Effective shader clock under the fused kernel: GRBM_GUI_ACTIVE / 8 / duration on a
# long dispatch (1 Mi frames ~ 1.3 ms), for the product build and the ablation builds.
export TMPDIR=/tmp
for lib in rtl-ws_amd/lib/librtlws_hip.so rtl-ws_amd/lib/variants/*/librtlws_hip.so; do
  tag=$(basename $(dirname $lib))
  OUT=clk_$tag; mkdir -p /tmp/$OUT
  RTLWS_HIP_LIB=$PWD/$lib rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/$OUT -- python3 bench.py --steps 30 --warmup 5 --sets 2 --frames 1048576 --no-cpu-baseline > /dev/null 2> /tmp/$OUT.err
  python3 - /tmp/$OUT $tag <<'PY'
import csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*_counter_collection.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'spectra_fused' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
rows = rows[5:]
clk = [float(r['Counter_Value']) / 8 / (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print('%-12s launches %d  avg duration %.1f us  effective clock %.3f GHz' % (tag, len(rows), sum(dur)/len(dur), sum(clk)/len(clk)))
PY
done
