#!/bin/bash
# batch-size x mapping sweep of the step kernel (runs on the GPU box).  usage: sweep_gpu.sh "sizes" "mappings"
SIZES=${1:-"1024 4096 16384 32768 65536 131072 262144"}
MAPS=${2:-"lane quad"}
for n in $SIZES; do
  for m in $MAPS; do
    steps=$(( 4000000 / n )); [ $steps -lt 100 ] && steps=100; [ $steps -gt 2000 ] && steps=2000
    python bench.py --envs-per-gpu $n --mapping $m --steps $steps --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('n=%7d %-5s  %8.1f us/step  kernel %8.1f us  %7.1f M env-steps/s  hbm %.2f%%' % (d['config']['envs_per_gpu'], '$m', d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']/1e6, 100*r['frac']))"
  done
done
