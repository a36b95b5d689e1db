#!/bin/bash
# Round-4 evidence run (one gpurun call): the driver's default line twice (default steps, --steps 20),
# the rocprofv3 kernel-trace + PMC passes of every default-line workload (tools/profile_gpu.sh), the
# drop-in latencies with and without the fused f64 kernel, the power / clock probe of the new kernels.
TAG=r04
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${TAG}_bench_default_line.json 2> gpurun_out/${TAG}_bench_default_line.err
echo "default line rc=$?"
python3 bench.py --steps 20 > gpurun_out/${TAG}_bench_default_line_steps20.json 2>> gpurun_out/${TAG}_bench_default_line.err
echo "steps20 line rc=$?"
for wl in batched_1024pt_64k_frames batched_1024pt_64k_frames_f64c_f32o batched_1024pt_64k_frames_f64 cic8_2048pt_f64 cic8_2048pt hann_4096pt_k8_db cic12_2048pt cic8_2048pt_f64c_f32o; do
  bash tools/profile_gpu.sh ${TAG}_$wl $wl > gpurun_out/prof_${TAG}_$wl.log 2>&1
  echo "profiled $wl"
done
python3 tools/dropin_latency.py > gpurun_out/${TAG}_dropin_latency.txt 2>&1
echo "== RTLWS_F64_FUSED=0 (row-per-workgroup f64 kernel)" >> gpurun_out/${TAG}_dropin_latency.txt
RTLWS_F64_FUSED=0 python3 tools/dropin_latency.py >> gpurun_out/${TAG}_dropin_latency.txt 2>&1
cat gpurun_out/${TAG}_dropin_latency.txt
rm -f gpurun_out/${TAG}_power_clocks.txt
bash tools/power_ab.sh gpurun_out/${TAG}_power_clocks.txt batched_1024pt_64k_frames_f64c_f32o 30000 product::
bash tools/power_ab.sh gpurun_out/${TAG}_power_clocks.txt cic8_2048pt_f64 30000 product::
bash tools/power_ab.sh gpurun_out/${TAG}_power_clocks.txt batched_1024pt_64k_frames 60000 product::
cat gpurun_out/${TAG}_power_clocks.txt
