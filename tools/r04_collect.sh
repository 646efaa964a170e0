#!/bin/bash
# condense gpurun_out/ of tools/r04_profiles.sh into profiles/r04/ and stamp profiles/traffic_index.json (runs on the GPU box from
# r04_profiles.sh, or in the build container on the merged gpurun_out/)
cd /root/repo
mkdir -p profiles/r04
for t in link_n4096 pair_n32768_yaw link_n4096_fs20_imu walking_n4096 quad_n4096 pair_n262144_yaw seq16_n4096 seq16_n32768_yaw; do
  [ -d gpurun_out/prof_r04_$t ] && python tools/summarize_profile.py gpurun_out/prof_r04_$t profiles/r04/$t > /dev/null
done
if [ "${1:-}" != profiles-only ]; then
  cp gpurun_out/r04/bench_*.json profiles/r04/ 2>/dev/null
  for f in po_step_rate parity_report rollout_demo closed_loop_demo map_sweep; do
    [ -f gpurun_out/r04/$f.txt ] && grep -v amdgpu.ids gpurun_out/r04/$f.txt > profiles/r04/$f.txt
  done
fi
python tools/update_traffic_index.py profiles/r04/link_n4096_pmc.json link 4096 4 33 --flops
python tools/update_traffic_index.py profiles/r04/quad_n4096_pmc.json quad 4096 4 33 --flops
python tools/update_traffic_index.py profiles/r04/pair_n32768_yaw_pmc.json pair 32768 4 33 --flops
python tools/update_traffic_index.py profiles/r04/link_n4096_fs20_imu_pmc.json link 4096 20 21 --flops
[ -f profiles/r04/pair_n262144_yaw_pmc.json ] && python tools/update_traffic_index.py profiles/r04/pair_n262144_yaw_pmc.json pair 262144 4 33
python tools/update_traffic_index.py profiles/r04/walking_n4096_pmc.json link 4096 4 33 --flops --suffix walking
