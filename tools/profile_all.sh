#!/bin/bash
# Round-3 evidence run (one gpurun call): for every default-line workload the rocprofv3
# --kernel-trace --stats pass of the driver's command plus the PMC passes (tools/profile_gpu.sh),
# then the clock / power probe of the 4096-point Hann workload (sclk for roofline.valu_issue_frac).
TAG=${1:-r03}
for wl in batched_1024pt_64k_frames hann_4096pt_k8_db cic8_2048pt cic12_2048pt batched_1024pt_64k_frames_f64; do
  bash tools/profile_gpu.sh ${TAG}_$wl $wl > gpurun_out/prof_${TAG}_$wl.log 2>&1
  echo "profiled $wl"
done
rm -f gpurun_out/${TAG}_power_hann.txt
bash tools/power_ab.sh gpurun_out/${TAG}_power_hann.txt hann_4096pt_k8_db 60000 product::
bash tools/power_ab.sh gpurun_out/${TAG}_power_hann.txt batched_1024pt_64k_frames 60000 product::
bash tools/power_ab.sh gpurun_out/${TAG}_power_hann.txt batched_1024pt_64k_frames_f64 30000 product::
cat gpurun_out/${TAG}_power_hann.txt
