#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > gpurun_out/ab_epi_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/ab_epi_tests.log
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 3
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "16384 32768" 3 --random-yaw
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "262144" 2 --random-yaw
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 2 --frame-skip 20 --obs-mode 1
