OUT=gpurun_out/r04_x1024_ablation_clocks.txt; : > $OUT
V=$PWD/rtl-ws_amd/lib/variants
run() { RTLWS_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --workload batched_1024pt_64k_frames_f64c_f32o --steps 3000 --no-cpu-baseline --no-extra 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s us %.2f frac %.4f sclk %.3f GHz  -> %.1f k shader cycles per launch' % ('$1', r['avg_launch_us'], r['frac'], r['sclk_ghz'], r['avg_launch_us']*r['sclk_ghz']))" >> $OUT; }
for rep in 1 2; do run product ""; run no_stores $V/x_nostore/librtlws_hip.so; run no_hbm $V/x_nomem/librtlws_hip.so; run no_lds $V/x_nolds/librtlws_hip.so; done
cat $OUT
