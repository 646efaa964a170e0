#!/bin/bash
# round 3: the two-legs-per-lane kernel after a change -- its parity tests, then config 3 / config 4-total / walking timings
set -u
cd /root/repo
T=${1:-pair1}
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_walking_gpu.py -m gpu -q -x --timeout 600 -k "pair or config or mappings or golden or estimator or snapshot" > gpurun_out/r03_${T}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r03_${T}_tests.log
[ $rc -ne 0 ] && exit $rc
for cfg in "cfg3:--steps 2000 --warmup 200 --envs-per-gpu 32768 --random-yaw" "cfg3b:--steps 2000 --warmup 200 --envs-per-gpu 32768 --random-yaw" "n20000:--steps 1000 --warmup 100 --envs-per-gpu 20000 --random-yaw" "cfg4one:--steps 300 --warmup 50 --envs-per-gpu 262144 --random-yaw" "walk32k:--steps 500 --warmup 50 --envs-per-gpu 32768 --walking"; do
  name=${cfg%%:*}; opts=${cfg#*:}
  python bench.py $opts --no-cpu-baseline > gpurun_out/r03_${T}_bench_${name}.json 2>gpurun_out/r03_${T}_bench_${name}.err
  python -c "import json; d=json.loads(open('gpurun_out/r03_${T}_bench_${name}.json').read().strip().splitlines()[-1]); print('${name}', round(d['value']/1e6,2), 'M  step_us', round(d['ms_per_step']*1e3,3), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,3), d['config']['mapping'])"
done
