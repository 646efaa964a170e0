#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_walking_gpu.py -m gpu -x -q > gpurun_out/ab_walkpair_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/ab_walkpair_tests.log
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "20000 32768 65536" 3 --walking
