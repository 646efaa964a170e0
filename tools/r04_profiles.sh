#!/bin/bash
# round 4 evidence: rocprofv3 kernel traces + PMC passes of bench.py for the BASELINE configs and the new many-steps-per-launch forms,
# the bench lines themselves, PO step rates, phase clocks, closed-loop demo.  Runs on the GPU box; results under gpurun_out/,
# condensed into profiles/r04/ (tools/r04_collect.sh) ON THE BOX so that the bench lines printed afterwards read profile_stale false.
cd /root/repo
mkdir -p gpurun_out/r04
tools/profile_gpu.sh r04_link_n4096
tools/profile_gpu.sh r04_pair_n32768_yaw --envs-per-gpu 32768 --random-yaw
tools/profile_gpu.sh r04_link_n4096_fs20_imu --frame-skip 20 --obs-mode 1
tools/profile_gpu.sh r04_walking_n4096 --walking
tools/profile_gpu.sh r04_quad_n4096 --mapping quad
tools/profile_gpu.sh r04_pair_n262144_yaw --envs-per-gpu 262144 --random-yaw --steps 100
tools/profile_gpu.sh r04_seq16_n4096 --seq 16
tools/profile_gpu.sh r04_seq16_n32768_yaw --envs-per-gpu 32768 --random-yaw --seq 16
bash tools/r04_collect.sh profiles-only
mkdir -p gpurun_out/r04_final && cp -r profiles/r04 profiles/traffic_index.json gpurun_out/r04_final/
