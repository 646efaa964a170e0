#!/bin/bash
cd /root/repo
export QUADGYM_LIB=tools/lib_phase.so
python tools/phase_times.py plain 32768 4 && python tools/phase_times.py walking 32768 4 && python tools/phase_times.py plain 4096 4
