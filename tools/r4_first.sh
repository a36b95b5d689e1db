#!/bin/bash
# round 4, first GPU pass: the new f64 fused instantiations (tests), then their rates
set -o pipefail
OUT=gpurun_out/r04_first
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_f64_fused_r4_gpu.py tests/test_f64_fused_gpu.py tests/test_v2_gpu.py tests/test_bench_gpu.py -x -q > $OUT.pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT.pytest.log
tail -5 $OUT.pytest.log
for wl in batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64 batched_1024pt_64k_frames cic8_2048pt_f64 cic8_2048pt_f64c_f32o cic8_2048pt cic12_2048pt_f64 hann_4096pt_k8_db_f64c_f32o hann_4096pt_k8_db_f64; do
  timeout -k 10 120 python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-extra 2>>$OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s %.4g %s  frac %.4f wall %.4f  us %.2f  parity %s' % (d['config']['workload'], d['value'], d['unit'], r['frac'], r['frac_wall'], r['avg_launch_us'], json.dumps(d['parity'])))" | tee -a $OUT.rates.txt
done
