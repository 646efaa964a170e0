#!/bin/bash
# round 3: in-kernel phase clock (development build, tools/phase_times.sh build) of the PO step at 32768 envs and the plain / walking steps
cd /root/repo
export QUADGYM_LIB=tools/lib_phase.so
python tools/phase_times.py po 32768 10 && python tools/phase_times.py walking 32768 10 && python tools/phase_times.py po 4096 10 && python tools/phase_times.py plain 4096 4 && python tools/phase_times.py po 16384 10
