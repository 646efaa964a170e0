#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_po_env.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for r in 1 2 3; do for L in tools/lib_base.so tools/lib_new.so; do echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 4096 10 1500 2>&1 | grep "PO walking"; done; done
for r in 1 2; do for L in tools/lib_base.so tools/lib_new.so; do echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 1024 10 1500 2>&1 | grep "PO walking"; done; done
