#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats plus
# separate PMC passes.  The trace pass profiles THE SAME COMMAND THE DRIVER RUNS
# (default steps / warm-up, so the >= 500 settle launches are in it);
# tools/summarize_profile.py averages only the timed launches (drops the first
# 500 dispatches of the dominant kernel).  The counter passes use fewer steps:
# counters are per launch.  Output under gpurun_out/prof_<tag>/; copy the
# summaries you want judged into profiles/.
set -e
TAG=${1:-r02}
WORKLOAD=${2:-batched_1024pt_64k_frames}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-extra --no-energy --no-clock-probe --workload $WORKLOAD"
# (counter collection serialises kernels: the probe wavefront of bench.py would hold the launches back)
PMC="python3 bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-extra --no-energy --no-clock-probe --workload $WORKLOAD"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PMC > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PMC > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $PMC > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $PMC > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $PMC > /dev/null 2> $OUT/pmc_grbm.err
# condense here (the per-dispatch traces are large; only the summaries travel back)
python3 tools/summarize_profile.py $OUT $OUT/summary_$WORKLOAD $WORKLOAD > $OUT/summary_$WORKLOAD.log 2>&1 || true
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*_counter_collection.csv" -delete
ls $OUT
