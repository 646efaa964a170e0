#!/bin/bash
# same-box A/B/.. of several library builds: interleaved rounds, kernel time per launch.  usage: abn_libs.sh "libs" "sizes" rounds [bench args]
LIBS=$1; SIZES=${2:-"4096"}; R=${3:-3}; shift 3
for n in $SIZES; do
  steps=$(( 6000000 / n )); [ $steps -lt 100 ] && steps=100; [ $steps -gt 2000 ] && steps=2000
  for r in $(seq 1 $R); do
    for L in $LIBS; do
      QUADGYM_LIB=$L python bench.py --envs-per-gpu $n --steps $steps --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d %-22s kernel %8.2f us  step %8.2f us' % (d['config']['envs_per_gpu'], '$L'.split('/')[-1], d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3))"
    done
  done
done
