#!/bin/bash
# interleaved same-box A/B of the quad and pair mappings: kernel time per launch.  usage: pair_ab.sh "sizes" rounds
SIZES=${1:-"4096 16384 32768 65536 262144"}; R=${2:-2}
for n in $SIZES; do
  steps=$(( 6000000 / n )); [ $steps -lt 100 ] && steps=100; [ $steps -gt 2000 ] && steps=2000
  for r in $(seq 1 $R); do for m in quad pair; do
    timeout -k 10 120 python bench.py --envs-per-gpu $n --steps $steps --warmup 50 --no-cpu-baseline --mapping $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d %-5s kernel %8.2f us  %8.1f M/s' % (d['config']['envs_per_gpu'], '$m', d['roofline']['kernel_ms']*1e3, d['value']/1e6))" | tee -a gpurun_out/pair_ab.txt
  done; done
done
