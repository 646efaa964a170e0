#!/bin/bash
# Builds a development copy of the library with the in-kernel phase clock (tools/lib_phase.so, next to the production build) and prints
# where one launch spends its time.  Build part runs anywhere (hipcc cross-compiles); the measurement needs the GPU box:
#   tools/phase_times.sh build ;  gpurun -- 'tools/phase_times.sh run'
cd /root/repo
if [ "${1:-build}" = build ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-missing-braces -Xarch_device -ffinite-math-only \
    -Xarch_device -fno-signed-zeros -Xarch_device -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -DQG_PHASE_TIMES -shared \
    -o tools/lib_phase.so quadruped-gym_amd/csrc/qg_capi.hip -ldl
else
  export QUADGYM_LIB=tools/lib_phase.so
  python tools/phase_times.py plain 4096 4 && python tools/phase_times_resident.py 4096 4 && python tools/phase_times.py walking 4096 4 && python tools/phase_times.py po 4096 10 \
    && python tools/phase_times.py plain 32768 4 && python tools/phase_times.py walking 32768 4 \
    && python tools/phase_times.py plain 12000 4 && python tools/phase_times.py walking 12000 4
fi
