#!/bin/bash
cd /root/repo
python -m pytest tests/test_walking_gpu.py tests/test_po_env.py tests/test_env_api.py -m gpu -q -x > gpurun_out/r02_t_walk.log 2>&1; echo "walk tests rc=$?"; tail -15 gpurun_out/r02_t_walk.log
for i in 1 2; do python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline kernel %8.2f us  step %8.2f us' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3))"; done
for n in 4096 32768; do for m in auto pair; do python bench.py --walking --envs-per-gpu $n --mapping $m --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walking n=$n mapping=$m  step %8.2f us  %7.1f M env-steps/s' % (d['ms_per_step']*1e3, d['value']/1e6))"; done; done
python tools/rollout_demo.py 2>&1 | tail -5
