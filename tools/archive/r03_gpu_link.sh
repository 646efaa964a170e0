#!/bin/bash
# round 3: parity of the rewritten one-link-per-lane kernel + its timing (long run, driver-style run, frame_skip 20, walking, PO)
set -u
cd /root/repo
mkdir -p gpurun_out
T=${1:-link1}
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_walking_gpu.py tests/test_po_env.py -m gpu -q -x --timeout 600 > gpurun_out/r03_${T}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r03_${T}_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline > gpurun_out/r03_${T}_bench_long.json 2>gpurun_out/r03_${T}_bench_long.err && cat gpurun_out/r03_${T}_bench_long.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', d['value'], d['ms_per_step'], d['roofline'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_${T}_bench_k20.json 2>gpurun_out/r03_${T}_bench_k20.err && cat gpurun_out/r03_${T}_bench_k20.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k20', d['value'], d['ms_per_step'])"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --frame-skip 20 --obs-mode 1 > gpurun_out/r03_${T}_bench_cfg5.json 2>/dev/null && python -c "import sys,json; d=json.loads(open('gpurun_out/r03_${T}_bench_cfg5.json').read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'])"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --walking > gpurun_out/r03_${T}_bench_walking.json 2>/dev/null && python -c "import sys,json; d=json.loads(open('gpurun_out/r03_${T}_bench_walking.json').read().strip().splitlines()[-1]); print('walking', d['value'], d['ms_per_step'])"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --generic-model > gpurun_out/r03_${T}_bench_generic.json 2>/dev/null && python -c "import sys,json; d=json.loads(open('gpurun_out/r03_${T}_bench_generic.json').read().strip().splitlines()[-1]); print('generic', d['value'], d['ms_per_step'])"
python tools/po_step_rate.py > gpurun_out/r03_${T}_po_step_rate.txt 2>&1; tail -6 gpurun_out/r03_${T}_po_step_rate.txt
