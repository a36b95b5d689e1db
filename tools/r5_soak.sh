#!/bin/bash
# round-5 closing soak (one gpurun call): the whole GPU suite twice, longer fuzz seeds, the > 4 GiB batch, smoke(),
# and the larger-launch / fill-and-drain check of the configs[3] kernels
OUT=gpurun_out/r05_soak.txt; : > $OUT
for rep in 1 2; do
  timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_soak_gputests_$rep.log 2>&1; echo "pytest -m gpu (pass $rep) rc $?: $(tail -1 gpurun_out/r05_soak_gputests_$rep.log)" >> $OUT
done
for s in 521 522 523; do timeout -k 10 200 python3 tests/tools/fuzz_parity.py $s 60 2>&1 | tail -1 >> $OUT; done
for s in 61 62 63 64; do timeout -k 10 200 python3 tests/tools/fuzz_parity_f64.py $s 60 2>&1 | tail -1 >> $OUT; done
timeout -k 10 300 python3 tests/tools/big_batch_check.py 2>&1 | tail -2 >> $OUT
python3 -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r.get('energy') or {}; print('%-24s %-14s us %.2f frac %.4f sclk %.3f GHz; %.1f mJ per launch at %.0f W' % (d['config']['workload'], '$1', r['avg_launch_us'], r['frac'], r.get('sclk_ghz', 0), e.get('mj_per_launch', 0), e.get('watts', 0)))"; }
for rep in 1 2; do for wl in cic8_2048pt cic8_2048pt_f64; do
  timeout -k 10 120 python3 bench.py --workload $wl --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | line "8192 spectra" >> $OUT
  timeout -k 10 120 python3 bench.py --workload $wl --steps 500 --no-cpu-baseline --no-extra --frames 32768 2>/dev/null | line "32768 spectra" >> $OUT
done; done
cat $OUT
