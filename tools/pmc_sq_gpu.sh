#!/bin/bash
# SQ counter pass only (issue-rate view) of bench.py on the GPU box.  Usage: tools/pmc_sq_gpu.sh <tag> [bench args] -> gpurun_out/prof_<tag>/pmc_sq
set -u
TAG=${1:-run}; shift || true
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline --wakeup-ms 0 "$@" > $OUT/pmc_sq.log 2>&1 && echo "pmc_sq ok $TAG"
