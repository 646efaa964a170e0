"""What the timing bracket of bench.py costs by itself: K step launches between two fences, with and without the two event records.
usage (GPU box): python tools/bracket_overhead.py ; HSA_ENABLE_INTERRUPT=0 python tools/bracket_overhead.py"""
import os, sys, time
import torch
sys.path.insert(0, "/root/repo")
from quadruped_gym_amd.sim import BatchedSim
from quadruped_gym_amd import _abi
dev = torch.device("cuda:0")
n = 4096
sim = BatchedSim(n)
sim.reset(seed=0, flags=0)
pool = [torch.rand((n, 12), device=dev) * 2 - 1 for _ in range(16)]
packed = [torch.empty((n, 35), device=dev) for _ in range(2)]
compute = torch.cuda.current_stream(dev)
step = sim.bind_step_packed(pool, packed, stream=compute)
for k in range(3000):
    step(k & 15, k & 1)
torch.cuda.synchronize()
def bracket(K, events=True):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events: ev0.record(compute)
    for k in range(K):
        step(k & 15, k & 1)
    if events: ev1.record(compute)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return dt * 1e6, (ev0.elapsed_time(ev1) * 1e3 if events else 0.0)
for K in (0, 1, 20, 100):
    for ev in (True, False):
        r = [bracket(K, ev) for _ in range(30)]
        w = sorted(x[0] for x in r)[len(r) // 2]; e = sorted(x[1] for x in r)[len(r) // 2]
        print(f"HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT','-')} K={K:4d} events={ev}: wall {w:8.1f} us  event span {e:8.1f} us  -> fixed {w - K * 13.65:6.1f} us")
