#!/bin/bash
# round 3: walking task layer after a change of the estimator -- its tests, then walking / PO step timings against a reference build
set -u
cd /root/repo
T=${1:-walk1}
timeout -k 10 900 python -m pytest tests/test_walking_gpu.py tests/test_po_env.py -m gpu -q -x --timeout 600 > gpurun_out/r03_${T}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r03_${T}_tests.log
[ $rc -ne 0 ] && exit $rc
for r in 1 2; do for L in tools/lib_base.so tools/lib_new.so; do
  for cfg in "4096:--steps 1000 --warmup 100" "32768:--steps 500 --warmup 50 --envs-per-gpu 32768" "16384:--steps 500 --warmup 50 --envs-per-gpu 16384"; do
    name=${cfg%%:*}; opts=${cfg#*:}
    QUADGYM_LIB=$L python bench.py $opts --walking --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L walking $name', round(d['roofline']['kernel_ms']*1e3,2),'us')"
  done
  echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 32768 10 600 2>&1 | grep "PO walking" | sed 's/PO walking step (one launch; frame_skip 10, window 10), //'
  echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 4096 10 1000 2>&1 | grep "PO walking" | sed 's/PO walking step (one launch; frame_skip 10, window 10), //'
done; done
