#!/bin/bash
# round 4: soak runs and the API fuzzer on the final build (walking reward without FMA contraction, unit_zero option, resident forms)
cd /root/repo
{
echo "== soak link 4096 x 20000"; python tools/soak_gpu.py 4096 20000
echo "== soak pair 32768 x 5000"; python tools/soak_gpu.py 32768 5000
echo "== soak PO link 4096 x 10000 fs10 (reference NaN)"; python tools/soak_po_gpu.py 4096 10000 10 1
echo "== soak PO link 4096 x 10000 fs10 (nan_direction=False)"; python tools/soak_po_gpu.py 4096 10000 10 0
echo "== soak PO pair 20000 x 3000 fs10 (nan_direction=False)"; python tools/soak_po_gpu.py 20000 3000 10 0
echo "== soak PO quad 9000 x 3000 fs4 (nan_direction=False)"; python tools/soak_po_gpu.py 9000 3000 4 0
echo "== resident soak 4096 x 60 s"; python tools/soak_resident_gpu.py 4096 60 0
echo "== resident soak 1000 x 30 s"; python tools/soak_resident_gpu.py 1000 30 1
echo "== fuzz 60 s"; python tools/fuzz_api_gpu.py 60
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_soak.txt
