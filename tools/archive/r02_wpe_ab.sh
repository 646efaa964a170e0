#!/bin/bash
# quad kernel register-cap A/B (QG_QUAD_WPE = waves per SIMD the build is capped for) + pair, kernel time per launch
cd /root/repo
for n in 16384 32768 49152 65536 131072 262144; do
  steps=$(( 6000000 / n )); [ $steps -lt 100 ] && steps=100; [ $steps -gt 1500 ] && steps=1500
  for r in 1 2; do
    for w in 2 3 4; do
      QG_QUAD_WPE=$w python bench.py --envs-per-gpu $n --steps $steps --warmup 50 --no-cpu-baseline --mapping quad 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d quad wpe=$w kernel %8.2f us  %6.1f M env-steps/s' % (d['config']['envs_per_gpu'], d['roofline']['kernel_ms']*1e3, d['value']/1e6))"
    done
    python bench.py --envs-per-gpu $n --steps $steps --warmup 50 --no-cpu-baseline --mapping pair 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d pair        kernel %8.2f us  %6.1f M env-steps/s' % (d['config']['envs_per_gpu'], d['roofline']['kernel_ms']*1e3, d['value']/1e6))"
  done
done
