#!/bin/bash
# round 3, VERDICT item 2: where the large-batch kernels lose their issue slots -- stall counters at 262 144 envs (pair, quad at two waves
# per SIMD) and, for reference, the headline kernel at 4096 envs and the pair kernel at 32 768 (one wave per SIMD)
cd /root/repo
tools/pmc_stall_gpu.sh r03_pair_n262144 --envs-per-gpu 262144 --random-yaw
tools/pmc_stall_gpu.sh r03_quad_n262144 --envs-per-gpu 262144 --random-yaw --mapping quad
tools/pmc_stall_gpu.sh r03_pair_n32768 --envs-per-gpu 32768 --random-yaw
tools/pmc_stall_gpu.sh r03_link_n4096
