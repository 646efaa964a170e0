#!/bin/bash
# kernel time per launch against frame_skip: the slope is the substep, the intercept the launch's fixed part.  usage: fs_sweep.sh "mappings" "sizes"
cd /root/repo
for m in $1; do for n in $2; do for fs in 1 2 4 8; do python bench.py --mapping $m --envs-per-gpu $n --frame-skip $fs --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m n=$n fs=$fs kernel %8.2f us' % (d['roofline']['kernel_ms']*1e3))"; done; done; done
