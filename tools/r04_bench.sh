#!/bin/bash
# round 4, second call (after the profiles of tools/r04_profiles.sh have been condensed and profiles/traffic_index.json stamped with the
# build id): the bench lines of every BASELINE config and of the opt-in forms, PO step rates, parity report, demos, the mapping sweep.
cd /root/repo
mkdir -p gpurun_out/r04
B=gpurun_out/r04
python bench.py > $B/bench_cfg2.json 2> $B/bench_cfg2.err; echo "cfg2 rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $B/bench_cfg2_driver_style.json 2>/dev/null; echo "cfg2 K=20 rc=$?"
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $B/bench_cfg2_long.json 2>/dev/null; echo "cfg2 long rc=$?"
python bench.py --seq 16 --no-cpu-baseline > $B/bench_cfg2_seq16.json 2>/dev/null; echo "seq16 rc=$?"
python bench.py --seq 64 --no-cpu-baseline > $B/bench_cfg2_seq64.json 2>/dev/null; echo "seq64 rc=$?"
python bench.py --resident ahead --no-cpu-baseline > $B/bench_cfg2_resident_ahead.json 2>/dev/null; echo "resident ahead rc=$?"
python bench.py --resident closed --no-cpu-baseline > $B/bench_cfg2_resident_closed.json 2>/dev/null; echo "resident closed rc=$?"
python bench.py --envs-per-gpu 32768 --random-yaw --steps 1000 --warmup 100 --cpu-seconds 3 > $B/bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python bench.py --envs-per-gpu 32768 --random-yaw --seq 16 --steps 992 --no-cpu-baseline > $B/bench_cfg3_seq16.json 2>/dev/null; echo "cfg3 seq rc=$?"
python bench.py --envs-per-gpu 16384 --steps 992 --no-cpu-baseline > $B/bench_n16384.json 2>/dev/null; echo "16384 rc=$?"
python bench.py --envs-per-gpu 16384 --seq 16 --steps 992 --no-cpu-baseline > $B/bench_n16384_seq16.json 2>/dev/null; echo "16384 seq rc=$?"
python bench.py --envs-per-gpu 262144 --random-yaw --seq 16 --steps 208 --warmup 16 --no-cpu-baseline > $B/bench_cfg4_total_one_gpu_seq16.json 2>/dev/null; echo "cfg4 seq rc=$?"
python bench.py --envs-per-gpu 262144 --random-yaw --steps 200 --warmup 20 --no-cpu-baseline > $B/bench_cfg4_total_one_gpu.json 2>/dev/null; echo "cfg4 rc=$?"
python bench.py --frame-skip 20 --obs-mode 1 --steps 1000 --warmup 100 --cpu-seconds 3 > $B/bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python bench.py --frame-skip 20 --obs-mode 1 --seq 16 --steps 1008 --no-cpu-baseline > $B/bench_cfg5_seq16.json 2>/dev/null; echo "cfg5 seq rc=$?"
python bench.py --walking --steps 1000 --warmup 100 --no-cpu-baseline > $B/bench_walking.json 2>/dev/null; echo "walk rc=$?"
python bench.py --walking --envs-per-gpu 32768 --steps 500 --warmup 50 --no-cpu-baseline > $B/bench_walking_n32768.json 2>/dev/null; echo "walk32k rc=$?"
python bench.py --generic-model --steps 1000 --warmup 100 --no-cpu-baseline > $B/bench_generic_model.json 2>/dev/null; echo "generic rc=$?"
python bench.py --force-gather --steps 200 --warmup 20 --no-cpu-baseline > $B/bench_one_rank_rccl_group.json 2>/dev/null; echo "1-rank group rc=$?"
: > $B/po_step_rate.txt
for n in 4096 16384 32768; do
  python tools/po_step_rate.py $n 10 1000 >> $B/po_step_rate.txt 2>&1
done
python tools/parity_report.py 4096 > $B/parity_report.txt 2>&1; echo "parity rc=$?"
python tools/rollout_demo.py > $B/rollout_demo.txt 2>&1
python tools/closed_loop_demo.py 4096 4000 > $B/closed_loop_demo.txt 2>&1
{ bash tools/map_sweep.sh "link quad" "1024 4096 5120 8192"; bash tools/map_sweep.sh "quad pair" "16384 20000 32768 49152 65536 262144"; } > $B/map_sweep.txt 2>&1 || true
bash tools/r04_collect.sh
mkdir -p gpurun_out/r04_final && cp -r profiles/r04 profiles/traffic_index.json gpurun_out/r04_final/
