#!/bin/bash
# condense gpurun_out/ of tools/r03_profiles.sh into profiles/r03/ and stamp profiles/traffic_index.json (runs on the GPU box from
# r03_profiles.sh, or in the build container on the merged gpurun_out/)
cd /root/repo
mkdir -p profiles/r03
for t in link_n4096 pair_n32768_yaw link_n4096_fs20_imu walking_n4096 quad_n4096 pair_n262144_yaw; do
  python tools/summarize_profile.py gpurun_out/prof_r03_$t profiles/r03/$t > /dev/null
done
[ "${1:-}" = profiles-only ] || cp gpurun_out/r03/bench_*.json gpurun_out/r03/po_step_rate.txt gpurun_out/r03/parity_report.txt gpurun_out/r03/rollout_demo.txt profiles/r03/ 2>/dev/null
[ "${1:-}" = profiles-only ] || grep -v amdgpu.ids gpurun_out/r03/po_step_rate.txt > profiles/r03/po_step_rate.txt
python tools/update_traffic_index.py profiles/r03/link_n4096_pmc.json link 4096 4 33 --flops
python tools/update_traffic_index.py profiles/r03/quad_n4096_pmc.json quad 4096 4 33 --flops
python tools/update_traffic_index.py profiles/r03/pair_n32768_yaw_pmc.json pair 32768 4 33 --flops
python tools/update_traffic_index.py profiles/r03/link_n4096_fs20_imu_pmc.json link 4096 20 21 --flops
[ -f profiles/r03/pair_n262144_yaw_pmc.json ] && python tools/update_traffic_index.py profiles/r03/pair_n262144_yaw_pmc.json pair 262144 4 33
python tools/update_traffic_index.py profiles/r03/walking_n4096_pmc.json link 4096 4 33 --flops --suffix walking
