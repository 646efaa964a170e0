#!/bin/bash
cd /root/repo
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 4
tools/abn_libs.sh "tools/lib_base.so tools/lib_new.so" "4096" 2 --frame-skip 20 --obs-mode 1
