#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03
tools/profile_gpu.sh r03_link_n4096
python bench.py > gpurun_out/r03/bench_cfg2.json 2> gpurun_out/r03/bench_cfg2.err; echo "cfg2 rc=$?"
