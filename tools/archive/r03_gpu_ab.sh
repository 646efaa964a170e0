#!/bin/bash
cd /root/repo
A=${1:-tools/lib_base.so}; B=${2:-tools/lib_new.so}
tools/abn_libs.sh "$A $B" "32768" 4 --random-yaw
tools/abn_libs.sh "$A $B" "16384" 3
tools/abn_libs.sh "$A $B" "49152 262144" 2 --random-yaw
tools/abn_libs.sh "$A $B" "4096" 2
