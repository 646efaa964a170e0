#!/bin/bash
# round 3: the whole GPU suite, then the timing set of tools/r03_gpu_link.sh
set -u
cd /root/repo
mkdir -p gpurun_out
T=${1:-full}
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r03_${T}_alltests.log 2>&1; rc=$?
echo "all tests rc=$rc"; tail -5 gpurun_out/r03_${T}_alltests.log
[ $rc -ne 0 ] && exit $rc
for cfg in "long:--steps 2000 --warmup 100" "k20:--steps 20 --warmup 5" "cfg5:--steps 1000 --warmup 100 --frame-skip 20 --obs-mode 1" "walking:--steps 1000 --warmup 100 --walking" \
           "cfg3:--steps 1000 --warmup 100 --envs-per-gpu 32768 --random-yaw" "cfg4one:--steps 300 --warmup 50 --envs-per-gpu 262144 --random-yaw" "walk32k:--steps 500 --warmup 50 --envs-per-gpu 32768 --walking"; do
  name=${cfg%%:*}; opts=${cfg#*:}
  python bench.py $opts --no-cpu-baseline > gpurun_out/r03_${T}_bench_${name}.json 2>gpurun_out/r03_${T}_bench_${name}.err
  python -c "import json; d=json.loads(open('gpurun_out/r03_${T}_bench_${name}.json').read().strip().splitlines()[-1]); print('${name}', round(d['value']/1e6,2), 'M  step_us', round(d['ms_per_step']*1e3,3), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,3), d['config']['mapping'])"
done
