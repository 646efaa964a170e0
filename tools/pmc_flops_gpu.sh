#!/bin/bash
# FP32 flop counters of the step kernel (SQ_INSTS_VALU_FLOPS_FP32 and the per-class instruction counts) in one SQ pass of bench.py
# on the GPU box.  Usage: tools/pmc_flops_gpu.sh <tag> [bench args] -> gpurun_out/prof_<tag>/pmc_flops
set -u
TAG=${1:-run}; shift || true
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_flops -- python3 /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline --wakeup-ms 0 "$@" > $OUT/pmc_flops.log 2>&1 && echo "pmc_flops ok $TAG"
