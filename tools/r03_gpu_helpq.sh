#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_walking_gpu.py -m gpu -x -q > gpurun_out/helpq_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/helpq_tests.log
for n in 8192 12000 16384; do for r in 1 2; do for h in 0 1; do
  QG_LINK_HELPERS=$h python bench.py --walking --envs-per-gpu $n --steps 800 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walking $n helpers=$h kernel %8.2f us  step %8.2f us  %s' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['config']['mapping'][:20]))"
done; done; done
