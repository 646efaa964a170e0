#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_walking_gpu.py tests/test_po_env.py -m gpu -x -q > gpurun_out/helpq_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/helpq_tests.log
for n in 8192 16384; do for r in 1 2; do for L in tools/lib_base.so tools/lib_new.so; do echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py $n 10 800 2>&1 | grep "PO walking"; done; done; done
