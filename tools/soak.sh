#!/bin/bash
# Round soak on the GPU box: the whole GPU suite, the randomised differential tests, the >4 GiB batch.
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_fuzz_soak.txt
: > $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest -m gpu rc $?: $(tail -1 gpurun_out/${TAG}_gputests.log)" >> $OUT
for s in 411 412 413; do timeout -k 10 200 python tests/tools/fuzz_parity.py $s 60 2>&1 | tail -1 >> $OUT; done
for s in 41 42 43 44; do timeout -k 10 200 python tests/tools/fuzz_parity_f64.py $s 60 2>&1 | tail -1 >> $OUT; done
timeout -k 10 300 python tests/tools/big_batch_check.py 2>&1 | tail -2 >> $OUT
cat $OUT
python -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
tail -3 $OUT
