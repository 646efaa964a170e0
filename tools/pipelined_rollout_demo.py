"""Closed-loop rollouts of 8192 robots with a 33-64-64-12 policy on the GPU, two ways ("batch sizes"):
  one handle   8192 envs in one batch (AUTO = one leg per lane): policy(8192) -> step(8192), one stream, one hipGraph;
  two handles  2 x 4096 envs (one link per lane each), each half with its own policy call on its own stream, each captured in its own
               hipGraph, replayed side by side: one half steps while the policy works on the other, and the two halves' step kernels
               share the SIMDs as first and second resident waves.
Random weights; plain QuadrupedEnv step (README reward set).  usage (GPU box): python tools/pipelined_rollout_demo.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
G = 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(33, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                             torch.nn.Linear(64, 12), torch.nn.Tanh()).to(dev)


def task():
    t = _abi.default_task()
    t.auto_reset, t.use_fall, t.fall_height = 1, 1, 0.05
    return t


class Half:
    def __init__(self, n, base):
        self.sim = BatchedSim(n, task=task(), env_index_base=base)
        self.sim.reset(seed=0)
        self.n = n
        self.acts = torch.zeros((n, 12), device=dev)
        self.rows = torch.zeros((n, 35), device=dev)
        self.stream = torch.cuda.Stream(dev)
        self.graph = None

    def one_step(self):
        with torch.no_grad():
            self.acts.copy_(policy(self.rows[:, :33]))
        self.sim.step_device_packed(self.acts, self.rows)

    def capture(self):
        with torch.cuda.stream(self.stream):
            for _ in range(3):
                self.one_step()
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            for _ in range(G):
                self.one_step()


def run(parts):
    halves, base = [], 0
    for p in parts:
        halves.append(Half(p, base)); base += p
    for h in halves:
        h.capture()
    def replay_all():
        for h in halves:
            with torch.cuda.stream(h.stream):    # a graph replays on the CURRENT stream: each half on its own, so that they overlap
                h.graph.replay()
    for _ in range(20):
        replay_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // G):
        replay_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = (steps // G) * G
    tot = sum(parts)
    ok = all(bool(torch.isfinite(h.rows).all()) for h in halves)
    print(f"{'+'.join(map(str, parts)):12s} envs ({[{_abi.MAP_LINK: 'link', _abi.MAP_QUAD: 'quad', _abi.MAP_PAIR: 'pair'}[h.sim.mapping] for h in halves]}): "
          f"{dt / k * 1e6:7.2f} us per closed-loop step of {tot} envs = {tot * k / dt / 1e6:7.1f} M env-steps/s  finite {ok}", flush=True)
    for h in halves:
        h.sim.close()


run((4096,))
run((8192,))
run((4096, 4096))
run((16384,))
run((4096, 4096, 4096, 4096))
