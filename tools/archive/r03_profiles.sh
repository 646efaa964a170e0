#!/bin/bash
# round 3 evidence: rocprofv3 kernel traces + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE, FP32 flop counters) of bench.py for the BASELINE
# configs, the bench lines themselves, PO step rates, phase clock.  Runs on the GPU box; results under gpurun_out/, condensed into
# profiles/r03/ by tools/r03_collect.sh afterwards.
cd /root/repo
mkdir -p gpurun_out/r03
# 1. profiles first, 2. condensed and stamped into profiles/traffic_index.json ON THE BOX, 3. the bench lines (which then report
# roofline.profile_stale false for this build), 4. everything copied to gpurun_out/r03_final/ for the way home
tools/profile_gpu.sh r03_link_n4096
tools/profile_gpu.sh r03_pair_n32768_yaw --envs-per-gpu 32768 --random-yaw
tools/profile_gpu.sh r03_link_n4096_fs20_imu --frame-skip 20 --obs-mode 1
tools/profile_gpu.sh r03_walking_n4096 --walking
tools/profile_gpu.sh r03_quad_n4096 --mapping quad
tools/profile_gpu.sh r03_pair_n262144_yaw --envs-per-gpu 262144 --random-yaw --steps 100
bash tools/r03_collect.sh profiles-only
python bench.py > gpurun_out/r03/bench_cfg2.json 2> gpurun_out/r03/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r03/bench_cfg2_long.json 2>/dev/null; echo "cfg2 long rc=$?"
python bench.py --envs-per-gpu 32768 --random-yaw --steps 1000 --warmup 100 --cpu-seconds 3 > gpurun_out/r03/bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python bench.py --envs-per-gpu 262144 --random-yaw --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03/bench_cfg4_total_one_gpu.json 2>/dev/null; echo "cfg4 rc=$?"
python bench.py --frame-skip 20 --obs-mode 1 --steps 1000 --warmup 100 --cpu-seconds 3 > gpurun_out/r03/bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python bench.py --walking --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r03/bench_walking.json 2>/dev/null; echo "walk rc=$?"
python bench.py --walking --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r03/bench_walking_n32768.json 2>/dev/null; echo "walk32k rc=$?"
python bench.py --generic-model --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r03/bench_generic_model.json 2>/dev/null; echo "generic rc=$?"
: > gpurun_out/r03/po_step_rate.txt
for n in 4096 16384 32768 65536; do
  python tools/po_step_rate.py $n 10 1000 >> gpurun_out/r03/po_step_rate.txt 2>&1
  QG_PO_UNFUSED=1 python tools/po_step_rate.py $n 10 1000 >> gpurun_out/r03/po_step_rate.txt 2>&1
done
python tools/parity_report.py 4096 > gpurun_out/r03/parity_report.txt 2>&1; echo "parity rc=$?"
python tools/rollout_demo.py > gpurun_out/r03/rollout_demo.txt 2>&1

bash tools/r03_collect.sh
mkdir -p gpurun_out/r03_final && cp -r profiles/r03 profiles/traffic_index.json gpurun_out/r03_final/
