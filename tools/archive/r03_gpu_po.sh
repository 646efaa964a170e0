#!/bin/bash
# round 3: the observation pack fused into the wave-level step kernels -- parity with the separate launch, then its timing
set -u
cd /root/repo
mkdir -p gpurun_out
T=${1:-po1}
timeout -k 10 900 python -m pytest tests/test_po_env.py -m gpu -q -x --timeout 600 > gpurun_out/r03_${T}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -8 gpurun_out/r03_${T}_tests.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r03_${T}_po_step_rate.txt
for n in 4096 8192 16384 20000 32768 40000 65536; do
  python tools/po_step_rate.py $n 10 600 >> gpurun_out/r03_${T}_po_step_rate.txt 2>&1
  QG_PO_UNFUSED=1 python tools/po_step_rate.py $n 10 600 >> gpurun_out/r03_${T}_po_step_rate.txt 2>&1
done
grep "PO walking" gpurun_out/r03_${T}_po_step_rate.txt
