#!/bin/bash
# quad vs pair (and a variant library passed as $1 under the name pair1) across batch sizes; kernel time per launch
for n in 18432 20480 24576 28672 32768 36864 40960 49152 65536 131072; do
  steps=$(( 3000000 / n )); [ $steps -lt 100 ] && steps=100
  for cfg in "quad:" "pair:" "pair:$1"; do
    m=${cfg%%:*}; L=${cfg#*:}
    [ "$cfg" = "pair:" ] || [ -n "$L" ] || [ "$m" = quad ] || continue
    QUADGYM_LIB=$L timeout -k 10 120 python bench.py --envs-per-gpu $n --steps $steps --warmup 50 --no-cpu-baseline --mapping $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%7d %-5s %-22s kernel %8.2f us  %8.1f M/s' % (d['config']['envs_per_gpu'], '$m', '$L'.split('/')[-1], d['roofline']['kernel_ms']*1e3, d['value']/1e6))" | tee -a gpurun_out/pair_sweep.txt
  done
done
