#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_walking_gpu.py tests/test_po_env.py -m gpu -x -q > gpurun_out/help_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/help_tests.log
for r in 1 2 3; do for h in 0 1; do
  QG_LINK_HELPERS=$h python bench.py --walking --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walking 4096 helpers=$h kernel %8.2f us  step %8.2f us' % (d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3))"
done; done
for r in 1 2; do for h in 0 1; do echo -n "helpers=$h "; QG_LINK_HELPERS=$h python tools/po_step_rate.py 4096 10 1500 2>&1 | grep "PO walking"; done; done
for h in 0 1; do echo -n "helpers=$h "; QG_LINK_HELPERS=$h python tools/po_step_rate.py 1024 10 1500 2>&1 | grep "PO walking"; done
