#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_walking_gpu.py tests/test_po_env.py -m gpu -x -q -k "link or golden or fresh or 4096 or config5 or hipgraph or state or reset" > gpurun_out/ab_link_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/ab_link_tests.log
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 4
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 2 --frame-skip 20 --obs-mode 1
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 2 --walking
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "1000" 2
