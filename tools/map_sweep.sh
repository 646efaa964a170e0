#!/bin/bash
# kernel time per launch of several work mappings over batch sizes.  usage: map_sweep.sh "mappings" "sizes" [bench args]
cd /root/repo
M=$1; S=$2; shift 2
for n in $S; do for m in $M; do python bench.py --mapping $m --envs-per-gpu $n --steps 1500 --warmup 150 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%6d %-5s kernel %8.2f us  %8.1f M env-steps/s' % ($n, '$m', d['roofline']['kernel_ms']*1e3, d['value']/1e6))"; done; done
