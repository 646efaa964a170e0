#!/bin/bash
cd /root/repo
tools/abn_libs.sh "tools/lib_base.so tools/lib_ctrlpro.so" "4096" 4
tools/abn_libs.sh "tools/lib_base.so tools/lib_ctrlpro.so" "32768" 3 --random-yaw
tools/abn_libs.sh "tools/lib_base.so tools/lib_ctrlpro.so" "16384" 2
