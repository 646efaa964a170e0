#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [extra bench args]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-run}; shift || true
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
(python3 -c "from quadruped_gym_amd import _abi; print(_abi.load_library().qg_build_id().decode())" > $OUT/build_id.txt)
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --steps 300 --warmup 30 --no-cpu-baseline --wakeup-ms 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 && echo "trace ok"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 && echo "pmc_sq ok"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 && echo "pmc_fetch ok"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 && echo "pmc_write ok"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_flops -- $BENCH > $OUT/pmc_flops.log 2>&1 && echo "pmc_flops ok"
