#!/bin/bash
# round 3: soak runs and the API fuzzer on the round's kernels (rewritten link kernel, PO fused into the pair / quad kernels)
cd /root/repo
{
echo "== soak link 4096 x 20000"; python tools/soak_gpu.py 4096 20000
echo "== soak link ragged 2000 x 5000"; python tools/soak_gpu.py 2000 5000
echo "== soak pair 32768 x 5000"; python tools/soak_gpu.py 32768 5000
echo "== soak quad 12000 x 5000"; python tools/soak_gpu.py 12000 5000
echo "== soak PO link 4096 x 10000 fs10"; python tools/soak_po_gpu.py 4096 10000 10
echo "== soak PO pair 20000 x 3000 fs10"; python tools/soak_po_gpu.py 20000 3000 10
echo "== soak PO quad 9000 x 3000 fs4"; python tools/soak_po_gpu.py 9000 3000 4
echo "== soak PO quad2 40000 x 1500 fs4"; python tools/soak_po_gpu.py 40000 1500 4
echo "== fuzz 90 s"; python tools/fuzz_api_gpu.py 90
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak.txt
