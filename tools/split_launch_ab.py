"""The 4097-env cliff, one cheap experiment: 4097 .. 8192 envs as TWO launches per env-step on two streams --
the one-link-per-lane kernel for the first 4096 envs (one wave on every SIMD) and a second kernel for the remainder (whose waves
become the SIMDs' second residents) -- against the ONE launch AUTO issues for the whole batch (one leg per lane).
usage (GPU box): python tools/split_launch_ab.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd import _abi  # noqa: E402
from quadruped_gym_amd.sim import BatchedSim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
NAME = {_abi.MAP_LINK: "link", _abi.MAP_QUAD: "quad", _abi.MAP_PAIR: "pair", _abi.MAP_LANE: "lane"}


def task():
    t = _abi.default_task()
    t.use_fall, t.fall_height, t.auto_reset = 1, 0.05, 1
    return t


def sim_of(n, mapping=_abi.MAP_AUTO, base=0):
    s = BatchedSim(n, task=task(), env_index_base=base)
    s.set_mapping(mapping)
    s.reset(seed=0)
    return s


def time_streams(jobs, steps):
    """jobs: [(sim, stream)]; every job steps `steps` times on its own stream; microseconds per env-step of the slowest stream."""
    bufs = []
    for s, st in jobs:
        acts = [torch.rand((s.n, 12), device=dev) * 2 - 1 for _ in range(8)]
        out = [torch.empty((s.n, 35), device=dev) for _ in range(2)]
        bufs.append((s.bind_step_packed(acts, out, stream=st), st))
    for warm in (200, steps):
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in bufs]
        for (fn, st), (e0, e1) in zip(bufs, ev):
            e0.record(st)
        for k in range(warm):
            for fn, st in bufs:
                fn(k & 7, k & 1)
        for (fn, st), (e0, e1) in zip(bufs, ev):
            e1.record(st)
        torch.cuda.synchronize()
    return [e0.elapsed_time(e1) / steps * 1e3 for e0, e1 in ev]


for n in (5120, 6144, 8192):
    whole = sim_of(n)
    t_whole = time_streams([(whole, torch.cuda.Stream(dev))], steps)[0]
    name = NAME[whole.mapping]
    whole.close()
    line = f"{n:6d} envs: one launch ({name}) {t_whole:6.2f} us"
    for rest_map in (_abi.MAP_QUAD, _abi.MAP_LINK):
        a, b = sim_of(4096, _abi.MAP_LINK), sim_of(n - 4096, rest_map, base=4096)
        ta, tb = time_streams([(a, torch.cuda.Stream(dev)), (b, torch.cuda.Stream(dev))], steps)
        line += f" | link 4096 + {NAME[b.mapping]} {n - 4096}: {ta:6.2f} / {tb:6.2f} us (slower stream {max(ta, tb):6.2f})"
        a.close(); b.close()
    print(line, flush=True)
