#!/bin/bash
# round 2 evidence: rocprofv3 kernel traces + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE, FP32 flop counters) of bench.py for the BASELINE
# configs, the bench lines themselves, the parity report.  Runs on the GPU box; results under gpurun_out/, condensed into profiles/r02/
# by tools/summarize_profile.py afterwards.
cd /root/repo
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench_cfg2.json 2> gpurun_out/r02/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --envs-per-gpu 32768 --random-yaw --steps 1000 --warmup 100 --cpu-seconds 3 > gpurun_out/r02/bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python bench.py --envs-per-gpu 32768 --random-yaw --joint-jitter 0.1 --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r02/bench_cfg3_jitter.json 2>/dev/null; echo "cfg3j rc=$?"
python bench.py --envs-per-gpu 262144 --random-yaw --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_cfg4_total_one_gpu.json 2>/dev/null; echo "cfg4 rc=$?"
python bench.py --frame-skip 20 --obs-mode 1 --steps 1000 --warmup 100 --cpu-seconds 3 > gpurun_out/r02/bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python bench.py --walking --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r02/bench_walking.json 2>/dev/null; echo "walk rc=$?"
python bench.py --walking --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench_walking_n32768.json 2>/dev/null; echo "walk32k rc=$?"
python bench.py --no-track-ctrl --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r02/bench_cfg2_no_ctrl_tracking.json 2>/dev/null; echo "noctrl rc=$?"
tools/profile_gpu.sh r02_link_n4096
tools/profile_gpu.sh r02_quad_n4096 --mapping quad
tools/profile_gpu.sh r02_pair_n32768_yaw --envs-per-gpu 32768 --random-yaw
tools/profile_gpu.sh r02_link_n4096_fs20_imu --frame-skip 20 --obs-mode 1
tools/profile_gpu.sh r02_walking_n4096 --walking
python bench.py --generic-model --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r02/bench_generic_model.json 2>/dev/null; echo "generic rc=$?"
python tools/po_step_rate.py 4096 10 2000 > gpurun_out/r02/po_step_rate.txt 2>&1
QG_PO_UNFUSED=1 python tools/po_step_rate.py 4096 10 2000 >> gpurun_out/r02/po_step_rate.txt 2>&1
python tools/parity_report.py 4096 > gpurun_out/r02/parity_report.txt 2>&1; echo "parity rc=$?"
python tools/rollout_demo.py > gpurun_out/r02/rollout_demo.txt 2>&1
