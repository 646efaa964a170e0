#!/bin/bash
# SQ stall / issue counters of the step kernel (two passes of eight SQ counters) on the GPU box.
# Usage: tools/pmc_stall_gpu.sh <tag> [bench args] -> gpurun_out/prof_<tag>/stall_{a,b} + gpurun_out/<tag>_stall.txt
# Reading (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAIT_ANY = wave parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = wave has
# an instruction but cannot issue it (dependency, pipe busy, arbitration lost to another wave), SQ_ACTIVE_INST_ANY = issuing;
# the three are disjoint and add up to SQ_WAVE_CYCLES (all in quad-cycles, summed over waves).
set -u
TAG=${1:-run}; shift || true
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline --wakeup-ms 0 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/stall_a -- $BENCH > $OUT/stall_a.log 2>&1 && echo "stall_a ok $TAG"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/stall_b -- $BENCH > $OUT/stall_b.log 2>&1 && echo "stall_b ok $TAG"
cd /root/repo
{
  echo "# $TAG: bench.py $*   (per-dispatch means of the step kernel; SQ cycle counters are in quad-cycles summed over waves)"
  for p in a b; do python tools/pmc_summary.py $(ls $OUT/stall_$p/*/*_counter_collection.csv | head -1) qg_step; done
} > gpurun_out/${TAG}_stall.txt 2>&1
cat gpurun_out/${TAG}_stall.txt
