#!/bin/bash
# same-box interleaved comparison of several library builds: kernel time per launch.  usage: abn_gpu.sh "libA libB ..." "sizes" rounds [extra bench args]
LIBS=$1; SIZES=${2:-"4096"}; R=${3:-3}; shift 3 || true
for n in $SIZES; do
  steps=$(( 3000000 / n )); [ $steps -lt 100 ] && steps=100; [ $steps -gt 2000 ] && steps=2000
  for r in $(seq 1 $R); do
    for L in $LIBS; do
      QUADGYM_LIB=$L timeout -k 10 120 python bench.py --envs-per-gpu $n --steps $steps --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d %-28s kernel %8.2f us' % (d['config']['envs_per_gpu'], '$L'.split('/')[-1], d['roofline']['kernel_ms']*1e3))"
    done
  done
done
