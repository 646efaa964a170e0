#!/bin/bash
# same-box A/B of library builds: kernel time per launch over batch sizes, each build in turn, two rounds.
# usage: ab_libs.sh "lib paths" "sizes" [bench args]
cd /root/repo
L=$1; S=$2; shift 2
for round in 1 2; do for n in $S; do for lib in $L; do QUADGYM_LIB=$lib python bench.py --envs-per-gpu $n --steps 1500 --warmup 150 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%6d %-28s kernel %8.2f us  step %8.2f us  %8.1f M env-steps/s' % ($n, '$lib'.split('/')[-1], d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value']/1e6), flush=True)"; done; done; done
