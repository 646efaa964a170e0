#!/bin/bash
cd /root/repo
for r in 1 2 3; do for L in tools/lib_base.so tools/lib_new.so; do echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 32768 10 800 2>&1 | grep "PO walking"; done; done
for r in 1 2; do for L in tools/lib_base.so tools/lib_new.so; do echo -n "$L "; QUADGYM_LIB=$L python tools/po_step_rate.py 65536 10 400 2>&1 | grep "PO walking"; done; done
