#!/bin/bash
# extra SQ counter passes over the default bench (issue/FIFO/ifetch side)
export TMPDIR=/tmp
OUT=/tmp/pmcx; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_ACCUM_PREV --kernel-trace --output-format csv -d $OUT/c -- $CMD > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('a','b','c'):
    fs = glob.glob('/tmp/pmcx/%s/*/*_counter_collection.csv' % d)
    if not fs: print(d, 'no output'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'spectra_fused' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print('%-32s %.4g' % (k, sum(v)/len(v)))
PY
