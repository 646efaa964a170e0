#!/bin/bash
# round 3: every work mapping around AUTO's boundaries on one box (final build)
cd /root/repo
{
tools/map_sweep.sh "link quad" "1024 4096 5120 6144 8192"
tools/map_sweep.sh "quad pair" "16384 20000 24576 32768 40000 49152 57344 65536 131072 262144" --random-yaw
tools/map_sweep.sh "lane" "4096 32768"
} 2>&1 | tee gpurun_out/r03_map_sweep.txt
