#!/bin/bash
# round 3: the tests added this round (and everything else in the files they live in)
set -u
cd /root/repo
mkdir -p gpurun_out
T=${1:-new1}
timeout -k 10 1100 python -m pytest tests/test_parity_gpu.py tests/test_walking_gpu.py tests/test_po_env.py tests/test_bench_contract.py tests/test_env_api.py -m gpu -q -x --timeout 900 --durations=12 > gpurun_out/r03_${T}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -30 gpurun_out/r03_${T}_tests.log
