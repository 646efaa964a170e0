#!/bin/bash
# round 2, first GPU call: full GPU test suite, default bench line, counters list, flop counters of the headline config
set -u
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r02_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r02_tests.log
tail -5 gpurun_out/r02_tests.log
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r02_bench_default.json
(cd /tmp && rocprofv3 -L > /root/repo/gpurun_out/r02_counters.txt 2>&1; grep -c . /root/repo/gpurun_out/r02_counters.txt)
tools/pmc_flops_gpu.sh r02_quad_n4096
python tools/pmc_summary.py $(ls gpurun_out/prof_r02_quad_n4096/pmc_flops/*/*_counter_collection.csv | head -1) qg_step > gpurun_out/r02_flops_quad_n4096.txt 2>&1
cat gpurun_out/r02_flops_quad_n4096.txt
